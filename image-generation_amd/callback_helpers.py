"""The reference's training driver with the reference's signature and side effects:
``execute_training(set_progress, model, n_epochs, qpu, n_latents, loss_data=None, example_image=None)`` of
/root/reference/src/utils/callback_helpers.py:144-221 (and ``create_model_files`` of :70-108).

Same loop, same calls into the model (``step``, ``generate_output``, ``generate_reconstucted_samples``,
``generate_loss_plot``), same files written (``generated_json/problem_details.json``, the per-epoch figure JSONs, the
spins of the first generated sample in ``assets/model_diagram/latent_qpu.json``), same four figures returned -- so
the Dash callbacks of the reference (demo_callbacks.py:634-650) can call it unchanged.  The epoch / batch loop itself
is :func:`image_generation_amd.training.execute_training`; the per-batch ``generate_model_diagram`` of the reference
(three extra forward passes and PNG writes per training step, only when ``GENERATE_NEW_MODEL_DIAGRAM`` and an example
image are given) is UI decoration and is not run (``example_image`` is accepted and ignored).

tests/golden/epoch_n64.* is the reference's own function run over the CPU oracle; tests/test_gpu_epoch.py replays it
through this module on the GPU.
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Callable, Optional

from . import training

MODEL_PATH = Path("models")
JSON_FILE_DIR = "generated_json"
PROBLEM_DETAILS_PATH = f"{JSON_FILE_DIR}/problem_details.json"
IMAGE_GEN_FILE_PREFIX = "generated_epoch_"
IMAGE_RECON_FILE_PREFIX = "reconstructed_epoch_"
LOSS_PREFIX = "loss_"
MODEL_DIAGRAM_PATH = "assets/model_diagram/"
LATENT_QPU_FILE = MODEL_DIAGRAM_PATH + "latent_qpu.json"


def _sharpen_output() -> bool:
    try:  # the reference keeps the switch in its UI configuration (/root/reference/demo_configs.py:61)
        from demo_configs import SHARPEN_OUTPUT  # type: ignore

        return bool(SHARPEN_OUTPUT)
    except Exception:
        return False


def _is_main(model) -> bool:
    is_main = getattr(model, "is_main_rank", None)
    return True if is_main is None else bool(is_main())


def create_model_files(model, file_name: str, qpu: str, n_latents: int, n_epochs: int, loss_data: dict) -> None:
    """/root/reference/src/utils/callback_helpers.py:70-108: ``models/<file_name>/{dvae.pth, grbm.pth, parameters.json,
    losses.json}``, ``loss_data`` written as given."""
    if not _is_main(model):  # data-parallel: one writer (save() is rank-local; every rank holds the same model)
        return
    model.save(file_path=MODEL_PATH / file_name)
    with open(MODEL_PATH / file_name / "parameters.json", "w") as f:
        json.dump({"n_latents": n_latents, "n_epochs": n_epochs, "prefactor": model.PREFACTOR, "qpu": qpu,
                   "num_read": model.NUM_READS, "loss_function": model.LOSS_FUNCTION, "image_size": model.IMAGE_SIZE,
                   "batch_size": model.BATCH_SIZE, "dateset_size": model.DATASET_SIZE, "random_seed": model.RANDOM_SEED}, f)
    with open(MODEL_PATH / file_name / "losses.json", "w") as f:
        json.dump(loss_data, f)


def execute_training(set_progress: Optional[Callable], model, n_epochs: int, qpu: str, n_latents: int,
                     loss_data: Optional[dict] = None, example_image=None):
    """Returns ``(fig_output, fig_reconstructed, fig_mse_loss, fig_total_loss)`` of the last epoch."""
    sharpen = _sharpen_output()
    os.makedirs(JSON_FILE_DIR, exist_ok=True)
    os.makedirs(MODEL_DIAGRAM_PATH, exist_ok=True)
    figs = [None]

    def end_of_epoch(epoch: int, report: dict) -> None:
        if not _is_main(model):
            # data-parallel: rank 0 writes the side files and draws the figures.  The other ranks still make the
            # generation DRAW (rank-local, no files): every rank's sampler counters -- the Philox sweep index of its
            # globally numbered chains -- then advance together, so W ranks keep sampling what one GPU with W x C
            # chains would.  The reconstruction preview consumes nothing (data.preview_batch).
            if hasattr(model, "generate_images"):
                model.generate_images(sharpen=sharpen)
            return
        with open(PROBLEM_DETAILS_PATH, "w") as f:
            json.dump(report, f)
        fig_output = model.generate_output(latent_qpu_file=LATENT_QPU_FILE, sharpen=sharpen,
                                           save_to_file=f"{JSON_FILE_DIR}/{IMAGE_GEN_FILE_PREFIX}{epoch + 1}.json")
        fig_reconstructed = model.generate_reconstucted_samples(
            sharpen=sharpen, save_to_file=f"{JSON_FILE_DIR}/{IMAGE_RECON_FILE_PREFIX}{epoch + 1}.json")
        fig_mse, fig_total = model.generate_loss_plot(
            save_to_file_mse=f"{JSON_FILE_DIR}/{LOSS_PREFIX}mse_{epoch + 1}.json",
            save_to_file_total=f"{JSON_FILE_DIR}/{LOSS_PREFIX}total_{epoch + 1}.json", old_loss_data=loss_data)
        figs[0] = (fig_output, fig_reconstructed, fig_mse, fig_total)

    training.execute_training(model, n_epochs, qpu=qpu, n_latents=n_latents, set_progress=set_progress,
                              on_epoch_end=end_of_epoch)
    return figs[0] if figs[0] is not None else (None, None, None, None)  # non-main ranks: unpackable
