"""End-to-end iteration rate of neat_amd.runner on a synthetic scene directory: batches from Dataset.__getitem__ through a DataLoader
(the reference's loop) against batches assembled on the device (datasets.DeviceBatches).  python scripts/runner_rate.py [res] [views]"""
import sys, time, tempfile, pathlib, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.runner import TrainRunner
from tests.test_runner import _toy_scene
from neat_amd.synth import hocon_text as _hocon
res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
views = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tmp = pathlib.Path(tempfile.mkdtemp())
_toy_scene(tmp / "data" / "abc" / "toy", res=res, n_views=views)
for hip_dataset in (False, True):
    model = dict(synth.ABC_NEAT_A_MODEL_CONF)
    model["hip_precision"] = "bf16"
    model["hip_sampler_sync_free"] = True
    conf = {"train": {"expname": "rate", "dataset_class": "datasets.blender_hawp_dataset.BlenderDataset",
                      "model_class": "model.networks.neat_wfr_rend_a.VolSDFNetwork", "loss_class": "model.networks.loss_wfr.VolSDFLoss",
                      "learning_rate": 5.0e-4, "num_pixels": 1024, "checkpoint_freq": 1000, "hip_dataset": hip_dataset},
            "loss": dict(synth.ABC_NEAT_A_LOSS_CONF), "dataset": {"data_dir": "abc/toy", "img_res": [res, res], "reverse_coordinate": True},
            "model": model}
    path = tmp / f"c{int(hip_dataset)}.conf"
    path.write_text(_hocon(conf))
    r = TrainRunner(str(path), nepochs=3, exps_folder=str(tmp / "exps"), data_root=str(tmp / "data"), log_freq=10 ** 6)
    r.save_checkpoints = lambda epoch: None
    r.run()                                   # epochs 0..3: every view's graph is captured on its second visit
    torch.cuda.synchronize()
    r.start_epoch, r.nepochs = 4, 4 + max(1, 60 // views) - 1
    t0 = time.perf_counter()
    r.run()
    torch.cuda.synchronize()
    n = (r.nepochs - r.start_epoch + 1) * views
    print(f"{res}x{res}, {views} views, 1024 rays/step, bf16, ErrorBoundSampler: hip_dataset={hip_dataset}: "
          f"{1e3 * (time.perf_counter() - t0) / n:.2f} ms per iteration ({n} iterations; replays {r.trainer.replays}, eager {r.trainer.eager_steps})", flush=True)
