"""GPU replay of tests/golden/epoch_n64.*: the reference's VERBATIM ``execute_training``
(/root/reference/src/utils/callback_helpers.py:144-221) over its verbatim ``ModelWrapper`` on the CPU oracle
(tests/golden/make_golden.py::epoch_fixture), against this package's ``callback_helpers.execute_training`` over the
MI355X ``ModelWrapper`` on the same batches, Gumbel noise and dropout masks: same progress calls, same files, same
report, losses within 1e-5, the same pictures and curves in the four figures."""
import json
import os

import numpy as np
import pytest
import torch

import gen
from image_generation_amd import callback_helpers
from image_generation_amd.model_wrapper import ModelWrapper

pytestmark = pytest.mark.gpu


def test_reference_training_driver_replay(tmp_path, golden_dir, monkeypatch):
    pytest.importorskip("plotly")
    fx = dict(np.load(os.path.join(golden_dir, "epoch_n64.npz")))
    meta = json.load(open(os.path.join(golden_dir, "epoch_n64.json")))
    n_epochs, spe = int(fx["n_epochs"]), int(fx["steps_per_epoch"])
    model = ModelWrapper("Advantage_system4", n_latents=64, training_parameter_file=os.path.join(golden_dir, "step_params.yaml"))
    B = model.BATCH_SIZE
    images = torch.from_numpy(gen.make_images(B * spe, seed=1313)).reshape(spe, B, 1, 32, 32)
    model.set_dataloader([(images[k], torch.zeros(B, dtype=torch.int64)) for k in range(spe)])
    model.train_init(n_epochs=n_epochs)
    model.noise_hook = lambda step: {
        "gumbels": torch.from_numpy(fx["gumbels_train"][step]),
        "dropout_masks": [torch.from_numpy(fx[f"masks{l}"][step].astype(np.float32)) for l in range(4)]}
    # the eval-mode forward inside generate_reconstucted_samples draws Gumbel noise too: replay the recorded draws
    evals = iter(fx["gumbels_eval"])
    orig = model.generate_reconstucted_samples

    def with_recorded_noise(*a, **k):
        model._dvae.inject_gumbels(torch.from_numpy(next(evals)).to(model._device))
        return orig(*a, **k)

    model.generate_reconstucted_samples = with_recorded_noise
    progress = []
    monkeypatch.chdir(tmp_path)
    figs = callback_helpers.execute_training(lambda p: progress.append(list(p)), model, n_epochs, "Advantage_system4", 64,
                                             loss_data=meta["old_loss_data"], example_image=None)
    fig_output, fig_recon, fig_mse, fig_total = figs
    # the driver's observable behaviour
    assert progress == meta["progress"]
    assert sorted(os.listdir(meta["json_file_dir"])) == meta["files"]
    assert (callback_helpers.JSON_FILE_DIR, callback_helpers.PROBLEM_DETAILS_PATH, callback_helpers.LATENT_QPU_FILE) == \
        (meta["json_file_dir"], meta["problem_details_path"], meta["latent_qpu_file"])
    assert json.load(open(meta["problem_details_path"])) == meta["details"]
    assert model.sampler.calls == int(fx["sampler_calls"])  # per step one draw (+1 on GRBM steps), one per generate_output
    # losses: 1e-5 relative over the whole run
    np.testing.assert_allclose(model.losses["mse_losses"], fx["mse"], rtol=1e-5)
    np.testing.assert_allclose(model.losses["dvae_losses"], fx["dvae"], rtol=1e-5)
    # loss figures: old data + this run, x = batch index
    np.testing.assert_allclose(np.asarray(fig_mse.data[0].y, dtype=np.float64), fx["curve_mse"], rtol=1e-5)
    np.testing.assert_allclose(np.asarray(fig_total.data[0].y, dtype=np.float64), fx["curve_total"], rtol=1e-5)
    assert list(fig_mse.data[0].x) == list(fx["curve_x"])
    assert fig_mse.layout.xaxis.title.text == meta["loss_fig_xaxis_title"] and fig_mse.layout.yaxis.title.text == meta["loss_fig_yaxis_title"]
    # pictures: the generated grid (sampler draw -> eval decoder) and the interleaved reconstructions, as 8-bit images
    assert fig_output.layout.margin.to_plotly_json() == meta["fig_layout_margin"]
    assert fig_output.layout.xaxis.showticklabels == meta["fig_xaxis_showticklabels"]
    assert sorted(k for k in fig_output.data[0].to_plotly_json() if k != "z") == meta["image_trace_keys"]
    for fig, key in ((fig_output, "img_output"), (fig_recon, "img_recon")):
        got, want = gen.figure_image(fig).astype(np.int32), fx[key].astype(np.int32)
        assert got.shape == want.shape
        diff = np.abs(got - want)
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (key, int(diff.max()), float((diff > 0).mean()))
    # the spins of the first generated sample (the sampler is bit-exact; the GRBM it draws from agrees to rounding)
    latent = np.asarray(json.load(open(meta["latent_qpu_file"])), dtype=np.float32)
    assert latent.shape == fx["latent"].shape and float((latent != fx["latent"]).mean()) <= 0.05
    saved = json.load(open(os.path.join(meta["json_file_dir"], meta["image_gen_prefix"] + "1.json")))
    assert saved["data"][0]["type"] == meta["saved_image_fig_trace_type"]
