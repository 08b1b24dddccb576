"""FOCAL pretraining loop (reference: train_utils/pretrain.py:21-107): per batch zero_grad -> loss -> backward ->
step -> loss.item().  Under a data-parallel launch (torchrun, or `train.py -gpu=0,1,...`; the process group is created by
params/params_util.py before the first HIP call) the batch is this rank's share of the global batch: the projected embeddings
are all-gathered and gradients all-reduced over RCCL (focal_amd/distributed.py); weights start identical on every rank
(broadcast from rank 0); the every-10-epochs branch fits the KNN estimator on the features of ALL ranks' training shares, then
rank 0 alone validates, logs and writes the checkpoints while the others wait at a barrier."""
import logging
import os
import sys

import numpy as np
import torch

from general_utils.time_utils import time_sync
from general_utils.weight_utils import freeze_patch_embedding
from train_utils.eval_functions import val_and_logging
from train_utils.knn import compute_knn
from train_utils.loss_calc_utils import calc_pretrain_loss
from train_utils.lr_scheduler import define_lr_scheduler
from train_utils.model_selection import init_pretrain_framework
from train_utils.optimizer import define_optimizer

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from focal_amd import distributed as fdist  # noqa: E402
from focal_amd import runtime  # noqa: E402
from focal_amd.graph_step import CapturedTrainStep  # noqa: E402


def pretrain(args, backbone_model, augmenter, train_dataloader, val_dataloader, test_dataloader, loss_func, num_batches):
    default_model = init_pretrain_framework(args, backbone_model)
    optimizer = define_optimizer(args, default_model.parameters())
    lr_scheduler = define_lr_scheduler(args, optimizer)
    default_model = freeze_patch_embedding(args, default_model)
    if getattr(args, "init_weight", None):
        default_model.backbone.load_state_dict(torch.load(args.init_weight, map_location="cpu"))
    fdist.broadcast_module(default_model)
    rank0 = fdist.rank() == 0

    logging.info("---------------------------Start Pretraining Classifier-------------------------------")
    start = time_sync()
    latest_weight = os.path.join(args.weight_folder, f"{args.dataset}_{args.model}_{args.stage}_latest.pt")
    best_weight = os.path.join(args.weight_folder, f"{args.dataset}_{args.model}_{args.stage}_best.pt")
    train_state = os.path.join(args.weight_folder, f"{args.dataset}_{args.model}_{args.stage}_train_state.pt")
    epochs = args.dataset_config[args.learn_framework]["pretrain_lr_scheduler"]["train_epochs"]
    if getattr(args, "epochs", None):
        epochs = args.epochs
    best_val_loss = np.inf
    start_epoch = 0
    resume = getattr(args, "resume", False) and os.path.exists(train_state) and os.path.exists(latest_weight)
    if resume:  # (the reference has no resume; weights + optimizer moments + step + dropout stream + epoch restore the run)
        default_model.backbone.load_state_dict(torch.load(latest_weight, map_location="cpu"))
        st = torch.load(train_state, map_location="cpu")
        start_epoch, best_val_loss = st["epoch"] + 1, st["best_val_loss"]
        for e in range(start_epoch):
            lr_scheduler.step(e)  # the replayed schedule owns the learning rate (the saved state holds the PREVIOUS epoch's)
        if st.get("rng") is not None:
            runtime.rng_state(args.device).copy_(st["rng"].to(args.device))
        default_model.backbone.arena()  # the optimizer state lives next to the parameter arena: build it, then restore the moments
        optimizer.load_train_state(st["optimizer"])
    windows = 0
    # the step body (zero_grad -> both views through the backbone -> loss -> backward -> AdamW) is replayed from a hipGraph once its
    # shapes have repeated; the views themselves are drawn eagerly every step (focal_amd/graph_step.py)
    graphed = CapturedTrainStep(default_model, loss_func, optimizer, enabled=not getattr(args, "no_graph", False))
    augmenter.static_views = graphed.enabled
    # Round 5: the two random views of a step are drawn ON THE DEVICE, inside the step (Augmenter.forward_random_pair: the reference's
    # `augmenter.forward("random", x)` x 2, loss_calc_utils.py:4-5) -- no host draw, no upload, nothing eager between two replays but
    # the copy of the batch into the step's input buffers.  -host_draws (FOCAL_HOST_DRAWS=1) keeps the host-drawn form.
    device_draws = augmenter.device_draws_supported() and not getattr(args, "host_draws", False) and os.environ.get("FOCAL_HOST_DRAWS") != "1"
    logging.info(f"random views: drawn on the {'device, inside the step' if device_draws else 'host'}")
    for epoch in range(start_epoch, epochs):
        default_model.train()
        for owner in (train_dataloader, getattr(train_dataloader, "batch_sampler", None)):
            if hasattr(owner, "set_epoch"):
                owner.set_epoch(epoch)  # the data-parallel shuffle follows the training epoch (also after -resume)
        train_loss_list = []
        epoch_t0, epoch_windows = time_sync(), 0
        pending = None
        for i, (time_loc_inputs, _) in enumerate(train_dataloader):
            # the views of step k are drawn (host: coin flips, warp tables, up to ~5 ms for a TimeWarp of the audio window) while the
            # GPU still runs step k - 1; only then is that step's loss read (the reference's per-step `loss.item()`, one step late
            # on the wall clock, same values): the static loss buffer of the captured step must be read before the next replay
            # (the batch crosses to the device once; both draws then see the same device tensors, which is also what pairs the two
            # views of a modality into the halves of one tensor, data_augmenter/Augmenter.py::_view_slot)
            time_loc_inputs, _ = augmenter.move_to_target_device(time_loc_inputs, None)
            if device_draws:
                if pending is not None:
                    train_loss_list.append(pending.item())
                pending = graphed.step_from_inputs(augmenter, time_loc_inputs)
            else:
                augmenter.begin_step()  # view 1 -> first half, view 2 -> second half of the static two-view buffers, whatever came before
                view1 = augmenter.forward("random", time_loc_inputs)
                view2 = augmenter.forward("random", time_loc_inputs)
                if pending is not None:
                    train_loss_list.append(pending.item())
                pending = graphed(view1, view2)
            n = next(iter(next(iter(time_loc_inputs.values())).values())).shape[0] * fdist.world()
            windows += n
            epoch_windows += n
        if pending is not None:
            train_loss_list.append(pending.item())
        if epoch % 10 == 0 or epoch == epochs - 1:
            dt = max(time_sync() - epoch_t0, 1e-9)
            logging.info(f"epoch {epoch}: {epoch_windows / dt:.1f} windows/s ({epoch_windows} windows in {dt:.3f} s; "
                         f"{graphed.replays} graph replays, {graphed.eager_steps} eager steps so far)")
        if epoch % 10 == 0:
            terms = loss_func.last_terms.tolist() if getattr(loss_func, "last_terms", None) is not None else []
            logging.info(f"epoch {epoch}: terms[shared,private,orth,rank,total]={terms}")
            # KNN estimator on the training features (every rank's share, gathered), then validation / test loss + accuracy
            # (reference :76-92) and the checkpoints on rank 0 only
            knn_estimator = compute_knn(args, default_model.backbone, augmenter, train_dataloader)
            if rank0:
                with fdist.local_only():
                    val_acc, val_loss = val_and_logging(args, epoch, default_model, augmenter, val_dataloader, test_dataloader,
                                                       loss_func, float(np.mean(train_loss_list)), estimator=knn_estimator)
                torch.save(default_model.backbone.state_dict(), latest_weight)
                if val_loss < best_val_loss:
                    best_val_loss = val_loss
                    torch.save(default_model.backbone.state_dict(), best_weight)
                torch.save({"epoch": epoch, "best_val_loss": best_val_loss, "optimizer": optimizer.train_state(),
                            "rng": runtime.rng_state(args.device).cpu()}, train_state)
            if fdist.is_dist():
                torch.distributed.barrier()
        lr_scheduler.step(epoch)
    end = time_sync()
    logging.info(f"Total processing time: {(end - start): .3f} s  ({windows / max(end - start, 1e-9):.1f} windows/s)")
    return default_model
