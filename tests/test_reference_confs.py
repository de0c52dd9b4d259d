"""Container-side pin of the conf layer: every conf the reference ships parses through neat_amd/conf.py,
builds the HIP model class, and `synth.ABC_NEAT_A_MODEL_CONF / _LOSS_CONF` -- which every golden fixture and bench.py call "the
abc-neat-a model" -- equal the `model` / `loss` blocks of the reference's own confs/abc-neat-a.conf key for key.  A silent drift of
either would invalidate every "this is abc-neat-a" claim.  /root/reference does not exist on the GPU box: skipped there."""
import glob
import os

import pytest

from neat_amd import conf as C
from neat_amd import synth

CONF_DIR = "/root/reference/code/confs"
pytestmark = pytest.mark.skipif(not os.path.isdir(CONF_DIR), reason="/root/reference is only present in the build container")


def _plain(node):
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [_plain(v) for v in node]
    if isinstance(node, bool) or node is None or isinstance(node, str):
        return node
    return float(node)


def _shipped():
    return sorted(glob.glob(os.path.join(CONF_DIR, "*.conf")) + glob.glob(os.path.join(CONF_DIR, "*", "*.conf")))


def test_all_shipped_confs_are_found():
    assert len(_shipped()) == 8 and os.path.join(CONF_DIR, "abc-neat-a.conf") in _shipped()


@pytest.mark.parametrize("path", _shipped() or ["-"], ids=lambda p: os.path.basename(p))
def test_shipped_conf_parses_and_builds_the_model(path):
    from neat_amd import networks
    from neat_amd.loss import VolSDFLoss
    tree = C.parse_file(path)
    for block in ("train", "dataset", "model", "loss"):
        assert isinstance(tree.get_config(block), C.ConfTree), (path, block)
    # the accessors the reference's trainer and model constructor use (volsdf_train.py:141-182, rend_a :258-315)
    for key in ("train.dataset_class", "train.model_class", "train.loss_class"):
        assert "." in tree.get_string(key)
    assert tree.get_float("train.learning_rate") > 0 and tree.get_int("train.num_pixels") > 0
    model = tree.get_config("model")
    assert model.get_list("implicit_network.dims") == [256] * 8 and model.get_list("implicit_network.skip_in") == [4]
    assert model.get_int("ray_sampler.N_samples") == 64 and model.get_int("ray_sampler.max_total_iters") == 5
    m = networks.VolSDFNetwork(conf=model)                       # no conf key of any shipped file is refused
    assert sum(p.numel() for p in m.parameters() if p is not m.latents) == 1219274 - 64 * 256
    assert m.latents.shape == (model.get_int("global_junctions.num_junctions"), 256)
    assert "hip_precision" not in model and m.handle().precision == 4      # the conf names no build: the default, fp16x3
    VolSDFLoss(**dict(tree.get_config("loss").items()))


def test_synth_conf_is_the_reference_abc_neat_a_conf():
    tree = C.parse_file(os.path.join(CONF_DIR, "abc-neat-a.conf"))
    assert _plain(tree.get_config("model")) == _plain(synth.ABC_NEAT_A_MODEL_CONF)
    assert _plain(tree.get_config("loss")) == _plain(synth.ABC_NEAT_A_LOSS_CONF)
