"""Training runner in the shape of the reference's `training/exp_runner.py` + `VolSDFTrainRunner`
(code/training/volsdf_train.py:66-410): same conf files, same loop (per-iteration re-sampling of the rays and
ExponentialLR step), same checkpoint layout (`checkpoints/{Model,Optimizer,Scheduler}Parameters/{epoch,latest}.pth` with
the reference's dict keys), so checkpoints move between the two code bases.  Visualisation, tensorboard, git logging and
the open3d dumps of the reference runner are out of scope.

    python -m neat_amd.runner --conf /path/to/confs/abc-neat-a.conf --data_root /path/to/data --nepoch 2000

Class paths in the conf that name the reference's dataset / model / loss are mapped to their neat_amd counterparts (the
conf may also name neat_amd classes directly, which is all the reference's own runner needs).

Data parallel (new: the reference pins one GPU, volsdf_train.py:130-131):  `python -m neat_amd.runner --gpus N ...` re-executes
itself under torch.distributed.run, one rank per GPU.  Every rank renders its OWN view per iteration (the model's forward is
hard-wired to one view: `[0]` indexing at rend_a :427-439) with train.num_pixels / N rays, seed 42 + rank for everything it
draws; the model starts from rank 0's weights; after backward ONE flat all-reduce averages the 1 219 274 gradients
(neat_amd/dp.py), then every rank takes the same Adam step.  Checkpoints and the log come from rank 0."""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

from . import conf as conf_mod
from . import dp, rend_util
from .general import get_class

CLASS_MAP = {
    "datasets.blender_hawp_dataset.BlenderDataset": "neat_amd.datasets.BlenderDataset",
    "datasets.scene_hawp_dataset.SceneDataset": "neat_amd.datasets.SceneDataset",
    "model.networks.neat_wfr_rend_a.VolSDFNetwork": "neat_amd.networks.VolSDFNetwork",
    "model.networks.loss_wfr.VolSDFLoss": "neat_amd.loss.VolSDFLoss",
}
SUBDIRS = ("ModelParameters", "OptimizerParameters", "SchedulerParameters")


class TrainRunner:
    def __init__(self, conf, nepochs, exps_folder="exps", expname="", scan_id=-1, data_root="../data", device="cuda:0",
                 timestamp=None, precision=None, log_freq=50, rank=0, world=1):
        self.conf = conf_mod.parse_file(conf) if isinstance(conf, str) else conf
        self.nepochs = nepochs
        self.device = torch.device(device)
        self.rank, self.world = rank, world
        self.expname = self.conf.get_string("train.expname") + expname
        if scan_id != -1:
            self.expname += f"/{scan_id}"
        self.timestamp = timestamp or time.strftime("%Y_%m_%d_%H_%M_%S")
        self.expdir = os.path.join(exps_folder, self.expname)
        self.checkpoints_path = os.path.join(self.expdir, self.timestamp, "checkpoints")
        if self.rank == 0:
            for sub in SUBDIRS:
                os.makedirs(os.path.join(self.checkpoints_path, sub), exist_ok=True)
        cls = lambda key: get_class(CLASS_MAP.get(self.conf.get_string(key), self.conf.get_string(key)))
        dataset_conf = dict(self.conf.get_config("dataset").items())
        if scan_id != -1:
            dataset_conf["scan_id"] = scan_id
        ds_cls = cls("train.dataset_class")
        if ds_cls.__module__.startswith("neat_amd"):
            dataset_conf["data_root"] = data_root
        self.train_dataset = ds_cls(**dataset_conf)
        gen = torch.Generator()
        gen.manual_seed(dp.rank_seed(42, self.rank))      # each rank walks the views in its own order (one view per rank and step)
        self.train_dataloader = torch.utils.data.DataLoader(self.train_dataset, batch_size=1, shuffle=True, generator=gen,
                                                            collate_fn=self.train_dataset.collate_fn)
        self.model = cls("train.model_class")(conf=self.conf.get_config("model")).to(self.device)
        if precision is not None and hasattr(self.model, "set_precision"):
            self.model.set_precision(precision)
        self.loss = cls("train.loss_class")(**dict(self.conf.get_config("loss").items()))
        self.lr = self.conf.get_float("train.learning_rate")
        if self.device.type == "cuda":
            from .optim import FlatAdam
            self.optimizer = FlatAdam(self.model.parameters(), lr=self.lr)
        else:
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.lr)
        decay_rate = self.conf.get_float("train.sched_decay_rate", default=0.1)
        decay_steps = self.nepochs * len(self.train_dataset)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, decay_rate ** (1.0 / decay_steps))
        self.num_pixels = dp.shard_rays(self.conf.get_int("train.num_pixels"), self.world)      # C4: 4096 rays / 8 ranks = 512 per rank
        self.bucket = dp.FlatGradBucket(self.model.parameters())
        if self.world > 1:                                 # every rank starts from rank 0's weights
            import torch.distributed as dist
            flat = getattr(self.optimizer, "flat_param", None)
            if flat is not None:
                dist.broadcast(flat, src=0)
            else:
                for p in self.model.parameters():
                    dist.broadcast(p.data, src=0)
        # the step itself: eager the first time a view comes up, from then on replayed from that view's HIP graph (train.Trainer).
        # New optional conf key train.hip_graphs (default: on for CUDA); the reference loop is the eager path.
        from .train import Trainer
        self.trainer = Trainer(device=self.device, parts=(self.model, self.loss, self.optimizer, self.scheduler, self.bucket))
        if self.device.type == "cuda" and self.conf.get_bool("train.hip_graphs", default=True):
            self.trainer.auto_capture = 1
        # batches assembled on the device from resident per-view maps (datasets.DeviceBatches) instead of Dataset.__getitem__ on the
        # host; new optional conf key train.hip_dataset (default: on for CUDA and a dataset that offers it).  Views come up in the
        # DataLoader's shuffle order either way.
        self.batches = None
        if (self.device.type == "cuda" and hasattr(self.train_dataset, "device_batches")
                and self.conf.get_bool("train.hip_dataset", default=True)):
            self.batches = self.train_dataset.device_batches(self.device)
            gen2 = torch.Generator()
            gen2.manual_seed(dp.rank_seed(42, self.rank))
            self.view_order = torch.utils.data.DataLoader(range(len(self.train_dataset)), batch_size=1, shuffle=True, generator=gen2)
        self.checkpoint_freq = self.conf.get_int("train.checkpoint_freq", default=100)
        self.start_epoch = 0
        self.log_freq = log_freq

    def load_checkpoints(self, checkpoints_dir, checkpoint="latest"):
        """Continue from a run of this runner or of the reference's (volsdf_train.py:187-207)."""
        state = torch.load(os.path.join(checkpoints_dir, SUBDIRS[0], f"{checkpoint}.pth"), map_location=self.device)
        self.model.load_state_dict(state["model_state_dict"], strict=False)
        self.start_epoch = state["epoch"]

    def save_checkpoints(self, epoch):
        # a replayed step keeps the line-loss NaN flag on the device (train.Trainer.check_nan): never write a checkpoint past it
        # (the reference stops at the NaN, loss_wfr.py:66-67)
        if self.device.type == "cuda":
            self.trainer.check_nan()
        if self.rank != 0:
            return
        payload = (("model_state_dict", self.model.state_dict()), ("optimizer_state_dict", self.optimizer.state_dict()),
                   ("scheduler_state_dict", self.scheduler.state_dict()))
        for sub, (key, sd) in zip(SUBDIRS, payload):
            for name in (str(epoch), "latest"):
                torch.save({"epoch": epoch, key: sd}, os.path.join(self.checkpoints_path, sub, f"{name}.pth"))

    def run(self):
        history = []
        epoch = self.start_epoch
        for epoch in range(self.start_epoch, self.nepochs + 1):
            if epoch % self.checkpoint_freq == 0:
                self.save_checkpoints(epoch)
            if self.batches is None:
                self.train_dataset.change_sampling_idx(self.num_pixels)
            self.model.train()
            for it, item in enumerate(self.train_dataloader if self.batches is None else self.view_order):
                if self.batches is None:
                    indices, model_input, ground_truth = item
                    for k in ("intrinsics", "uv", "pose", "uv_proj"):
                        model_input[k] = model_input[k].to(self.device)
                else:
                    indices, model_input, ground_truth = self.batches.batch(int(item), self.num_pixels)
                # forward, loss, backward, [ONE flat all-reduce of all gradients when world > 1], Adam, scheduler
                outputs, losses = self.trainer.step(model_input, ground_truth)
                if self.batches is None:
                    self.train_dataset.change_sampling_idx(self.num_pixels)      # (a full randperm of the image per step: only its length is used)
                if (it + 1) % self.log_freq == 0 or it + 1 == len(self.train_dataloader):
                    if self.device.type == "cuda":
                        self.trainer.check_nan()            # (one sync per log interval; eager steps raise on their own)
                    with torch.no_grad():
                        psnr = rend_util.get_psnr(outputs["rgb_values"], ground_truth["rgb"].to(self.device).reshape(-1, 3))
                    history.append((epoch, it, float(losses["loss"].detach()), float(psnr)))
                    if self.rank == 0:
                        print(f"{self.expname}/{self.timestamp} [{epoch}] ({it}/{len(self.train_dataloader)}): "
                              f"loss = {history[-1][2]:.4f}, psnr = {history[-1][3]:.3f}", flush=True)
        self.save_checkpoints(epoch)
        return history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--conf", required=True)
    ap.add_argument("--nepoch", type=int, default=2000)
    ap.add_argument("--expname", default="")
    ap.add_argument("--exps_folder", default="exps")
    ap.add_argument("--scan_id", type=int, default=-1)
    ap.add_argument("--data_root", default="../data")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3", "fp16", "fp16x3"], default=None)
    ap.add_argument("--is_continue", default=None, help="checkpoints directory of the run to continue")
    ap.add_argument("--checkpoint", default="latest")
    ap.add_argument("--gpus", type=int, default=1, help="data-parallel ranks (one per GPU); > 1 re-executes under torch.distributed.run")
    ap.add_argument("--timestamp", default=None)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        stamp = args.timestamp or time.strftime("%Y_%m_%d_%H_%M_%S")      # one run directory for all ranks
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "neat_amd.runner"] + sys.argv[1:] + ["--timestamp", stamp]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank, world, local = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    seed = dp.rank_seed(42, rank)      # exp_runner.py:36,49-51 seeds torch / random / numpy with 42; rank r uses 42 + r
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    local = local % max(torch.cuda.device_count(), 1)      # (more ranks than GPUs only under NEAT_DIST_BACKEND=gloo: functional checks)
    if torch.cuda.is_available():
        torch.cuda.set_device(local)      # the ctypes launches take torch.cuda.current_stream(): it must be this rank's device, whatever the backend
    runner = TrainRunner(args.conf, args.nepoch, args.exps_folder, args.expname, args.scan_id, args.data_root, device=f"cuda:{local}",
                         timestamp=args.timestamp, precision=args.precision, rank=rank, world=world)
    if args.is_continue:
        runner.load_checkpoints(args.is_continue, args.checkpoint)
    runner.run()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
