"""Epoch / batch driver of the training path: the compute side of
/root/reference/src/utils/callback_helpers.py:144-221 (``execute_training``) and :70-108
(``create_model_files``), without the Dash / plotly coupling.

Same loop (``for epoch: for i, batch in enumerate(model._dataloader): model.step(batch, epoch)``), same
per-epoch report fields (``problem_details.json``), same side files (``parameters.json`` with the
reference's keys, including its ``dateset_size`` spelling, and ``losses.json``).  The per-batch
``generate_model_diagram`` call of the reference (three extra forward passes and PNG writes per step,
:181-182) is UI work and is not part of the timed path.
"""
from __future__ import annotations

import json
import time
from pathlib import Path
from typing import Callable, Optional

import torch


def execute_training(model, n_epochs: int, qpu: Optional[str] = None, n_latents: Optional[int] = None,
                     set_progress: Optional[Callable] = None, details_path: Optional[str] = None, verbose: bool = True,
                     on_epoch_end: Optional[Callable] = None):
    """Runs ``n_epochs`` over ``model._dataloader``; ``model.train_init(n_epochs)`` must have been called.
    Returns a list with one dict per epoch (the fields the reference prints / dumps).  ``on_epoch_end(epoch, report)``
    is called after each epoch with the reference's seven report fields (callback_helpers.execute_training hangs the
    per-epoch figure generation there)."""
    reports = []
    for epoch in range(n_epochs):
        start = time.perf_counter()
        total = len(model._dataloader)
        mse_loss = None
        for i, batch in enumerate(model._dataloader):
            if set_progress is not None:
                set_progress((str(total * epoch + i), str(total * n_epochs)))
            mse_loss = model.step(batch, epoch)
        if model._device.type == "cuda":
            torch.cuda.synchronize(model._device)
        # data-parallel: the one point of an epoch every rank reaches outside the step -- BatchNorm running statistics
        # become rank 0's here, so that the rank-local save() / generate_*() behind it see one model on every rank
        if hasattr(model, "sync_buffers"):
            model.sync_buffers()
        lr_dvae = float(model._tpar["dvae_lr_schedule"][model._tpar["opt_step"]])
        lr_grbm = float(model._tpar["grbm_lr_schedule"][model._tpar["opt_step"]])
        minutes = (time.perf_counter() - start) / 60
        report = {
            "QPU": qpu or model.qpu,
            "Epoch": f"{epoch + 1}/{n_epochs}",
            "Batch Size": model.BATCH_SIZE,
            "Latents": n_latents or model.n_latents,
            "Learning rate DVAE": f"{lr_dvae:.3E}",
            "Learning rate GRBM": f"{lr_grbm:.3E}",
            "Mean Squared Error Loss": f"{float(mse_loss):.4f}",
        }
        if verbose:
            print(f"Epoch {epoch + 1}/{n_epochs} - MSE Loss: {float(mse_loss):.4f} - Learning rate DVAE: {lr_dvae:.3E} "
                  f"Learning rate GRBM: {lr_grbm:.3E} Time: {minutes:.2f} mins. "
                  f"({total * model.BATCH_SIZE / (minutes * 60):.0f} images/s)")
        is_main = getattr(model, "is_main_rank", None)
        if details_path and (is_main is None or is_main()):
            with open(details_path, "w") as f:
                json.dump(report, f)
        if on_epoch_end is not None:
            on_epoch_end(epoch, dict(report))
        reports.append(dict(report, minutes=minutes))
    return reports


def create_model_files(model, model_dir, n_epochs: int, loss_data: Optional[dict] = None):
    """``dvae.pth`` + ``grbm.pth`` + ``parameters.json`` + ``losses.json`` in the reference's format
    (/root/reference/src/utils/callback_helpers.py:70-108)."""
    model_dir = Path(model_dir)
    is_main = getattr(model, "is_main_rank", None)
    if is_main is not None and not is_main():  # data-parallel: one writer
        return
    model.save(model_dir)
    with open(model_dir / "parameters.json", "w") as f:
        json.dump(
            {
                "n_latents": model.n_latents,
                "n_epochs": n_epochs,
                "prefactor": model.PREFACTOR,
                "qpu": model.qpu,
                "num_read": model.NUM_READS,
                "loss_function": model.LOSS_FUNCTION,
                "image_size": model.IMAGE_SIZE,
                "batch_size": model.BATCH_SIZE,
                "dateset_size": model.DATASET_SIZE,
                "random_seed": model.RANDOM_SEED,
            },
            f,
        )
    losses = {k: [float(v) for v in vals] for k, vals in model.losses.items()}
    if loss_data:
        losses = {k: list(loss_data.get(k, [])) + losses[k] for k in losses}
    with open(model_dir / "losses.json", "w") as f:
        json.dump(losses, f)
