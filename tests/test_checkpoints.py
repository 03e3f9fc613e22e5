"""The six checkpoints shipped with the reference (models/*/{dvae,grbm}.pth) load UNCHANGED into the
MI355X-native modules (strict state_dict match: SURVEY.md App. B).  Needs /root/reference (build container only);
the checkpoints are 28 MB of third-party data and are not copied into this repository."""
import glob
import os

import pytest
import torch

from image_generation_amd.model_wrapper import ModelWrapper
from image_generation_amd.modules import Decoder, Encoder
from image_generation_amd.plugin import DiscreteVariationalAutoencoder, GraphRestrictedBoltzmannMachine
from oracle import nets

MODELS = sorted(glob.glob("/root/reference/models/*/dvae.pth"))
pytestmark = pytest.mark.skipif(not MODELS, reason="reference checkpoints not available on this box")


@pytest.mark.parametrize("path", MODELS)
def test_reference_checkpoint_loads_strictly(path):
    folder = os.path.dirname(path)
    dvae_sd = torch.load(path, weights_only=True)
    grbm_sd = torch.load(os.path.join(folder, "grbm.pth"), weights_only=True)
    n = dvae_sd["_encoder.projection.weight"].numel() and dvae_sd["_decoder.increase_latent_dim.weight"].shape[1]
    dvae = DiscreteVariationalAutoencoder(Encoder(n), Decoder(n))
    missing, unexpected = dvae.load_state_dict(dvae_sd, strict=True)
    assert not missing and not unexpected
    assert list(dvae.state_dict().keys()) == list(dvae_sd.keys())
    ei, ej = grbm_sd["_edge_idx_i"].tolist(), grbm_sd["_edge_idx_j"].tolist()
    grbm = GraphRestrictedBoltzmannMachine(range(n), zip(ei, ej))
    grbm.load_state_dict(grbm_sd, strict=True)
    assert list(grbm.state_dict().keys()) == list(grbm_sd.keys())
    # the real-QPU sub-graph is a valid sampler graph: proper colouring, CSR consistent with the edge list
    plan = grbm.plan
    assert plan.n == n and plan.n_edges == len(ei) and plan.n_colours <= 6
    # and the loaded weights mean the same network: oracle forward (stock torch ops) == the stock sub-modules
    x = (torch.rand(3, 1, 32, 32) < 0.13).float()
    dvae.eval()
    enc_sd = {k[len("_encoder."):]: v for k, v in dvae_sd.items() if k.startswith("_encoder.")}
    with torch.no_grad():
        want = dvae.encoder.projection(dvae.encoder.flatten_last_two_dims(dvae.encoder.conv(x))).flatten(1)
        got = nets.encoder_forward({k: v.clone() for k, v in enc_sd.items()}, x, training=False)
    assert torch.allclose(got, want, atol=1e-5)


def test_model_wrapper_load_rebuilds_sampler_on_checkpoint_graph(tmp_path):
    """ModelWrapper.load accepts a checkpoint trained on another (real-QPU) sub-graph: the GRBM and the local
    sampler are rebuilt on the checkpoint's own edge list (the reference needs the same live QPU for that)."""
    src = os.path.dirname(MODELS[0])
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_params.yaml")
    m = ModelWrapper("Advantage2_system1", n_latents=256, training_parameter_file=golden)
    m.set_dataloader([(torch.zeros(8, 1, 32, 32), torch.zeros(8))])
    from pathlib import Path

    m.load(Path(src))
    want = torch.load(os.path.join(src, "grbm.pth"), weights_only=True)
    assert torch.equal(m._grbm._quadratic.detach().cpu(), want["_quadratic"])
    assert m.sampler.plan.n_edges == want["_quadratic"].numel()
    assert torch.equal(m._dvae.state_dict()["_decoder.convtrans.20.weight"].cpu(),
                       torch.load(os.path.join(src, "dvae.pth"), weights_only=True)["_decoder.convtrans.20.weight"])
