"""Do the 16-bit builds TRAIN like the fp32 build?  The same synthetic scene directory, the same initial weights and the same random
streams, `neat_amd.runner` for N iterations per precision; prints the mean loss / rgb PSNR over windows of the run.
    python scripts/train_curves.py [iterations] [rays] [precisions, comma separated]"""
import sys, tempfile, pathlib, random
import numpy as np, torch
sys.path.insert(0, '.')
from neat_amd import synth, rend_util
from neat_amd.runner import TrainRunner
from tests.test_runner import _toy_scene
from neat_amd.synth import hocon_text as _hocon
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 512
views, res = 4, 128
tmp = pathlib.Path(tempfile.mkdtemp())
_toy_scene(tmp / "data" / "abc" / "toy", res=res, n_views=views)
rows = {}
precs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["fp32", "bf16x3", "bf16", "fp16"]
for prec in precs:
    torch.manual_seed(42); np.random.seed(42); random.seed(42)
    model = dict(synth.ABC_NEAT_A_MODEL_CONF)
    model["hip_precision"] = prec
    conf = {"train": {"expname": "curves", "dataset_class": "datasets.blender_hawp_dataset.BlenderDataset",
                      "model_class": "model.networks.neat_wfr_rend_a.VolSDFNetwork", "loss_class": "model.networks.loss_wfr.VolSDFLoss",
                      "learning_rate": 5.0e-4, "num_pixels": rays, "checkpoint_freq": 10 ** 6},
            "loss": dict(synth.ABC_NEAT_A_LOSS_CONF), "dataset": {"data_dir": "abc/toy", "img_res": [res, res], "reverse_coordinate": True},
            "model": model}
    path = tmp / f"{prec}.conf"
    path.write_text(_hocon(conf))
    r = TrainRunner(str(path), nepochs=iters // views - 1, exps_folder=str(tmp / "exps"), data_root=str(tmp / "data"), log_freq=1)
    r.save_checkpoints = lambda epoch: None
    hist = r.run()
    loss = np.array([h[2] for h in hist]); psnr = np.array([h[3] for h in hist])
    assert np.isfinite(loss).all(), prec
    w = len(loss) // 6
    rows[prec] = (loss, psnr)
    print(f"{prec:7s} loss per sixth of the run: " + " ".join(f"{loss[i * w:(i + 1) * w].mean():.4f}" for i in range(6)) +
          "   psnr: " + " ".join(f"{psnr[i * w:(i + 1) * w].mean():.2f}" for i in range(6)) +
          f"   (replays {r.trainer.replays}, eager {r.trainer.eager_steps})", flush=True)
ref = rows[precs[0]]
for prec in precs[1:]:
    w = len(ref[0]) // 6
    d = abs(rows[prec][0][-w:].mean() - ref[0][-w:].mean()) / ref[0][-w:].mean()
    print(f"{prec}: final-window loss differs from {precs[0]}'s by {100 * d:.2f} %; psnr {rows[prec][1][-w:].mean() - ref[1][-w:].mean():+.2f} dB")
