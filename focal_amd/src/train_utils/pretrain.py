"""FOCAL pretraining loop (reference: train_utils/pretrain.py:21-107): per batch zero_grad -> loss -> backward ->
step -> loss.item().  Under torchrun (WORLD_SIZE > 1) the batch is the local shard of a data-parallel job: the
projected embeddings are all-gathered and gradients all-reduced over RCCL (focal_amd/distributed.py)."""
import logging
import os

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


def pretrain(args, backbone_model, augmenter, train_dataloader, val_dataloader, test_dataloader, loss_func, num_batches):
    default_model = init_pretrain_framework(args, backbone_model)
    optimizer = define_optimizer(args, default_model.parameters())
    lr_scheduler = define_lr_scheduler(args, optimizer)
    default_model = freeze_patch_embedding(args, default_model)

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
    if resume:  # (the reference has no resume; weights + optimizer moments + step + epoch restore the run exactly)
        default_model.backbone.load_state_dict(torch.load(latest_weight, map_location="cpu"))
        st = torch.load(train_state, map_location="cpu")
        start_epoch, best_val_loss = st["epoch"] + 1, st["best_val_loss"]
        for e in range(start_epoch):
            lr_scheduler.step(e)
    windows = 0
    for epoch in range(start_epoch, epochs):
        default_model.train()
        train_loss_list = []
        for i, (time_loc_inputs, _) in enumerate(train_dataloader):
            optimizer.zero_grad()
            loss = calc_pretrain_loss(args, default_model, augmenter, loss_func, time_loc_inputs)
            loss.backward()
            if resume and i == 0 and epoch == start_epoch:
                optimizer.load_train_state(st["optimizer"])  # the arena exists once a backward has run
            optimizer.step()
            train_loss_list.append(loss.item())
            windows += args.batch_size
        if epoch % 10 == 0:
            terms = loss_func.last_terms.tolist() if getattr(loss_func, "last_terms", None) is not None else []
            logging.info(f"epoch {epoch}: terms[shared,private,orth,rank,total]={terms}")
            # KNN estimator on the training features, then validation / test loss + accuracy (reference :76-92)
            knn_estimator = compute_knn(args, default_model.backbone, augmenter, train_dataloader)
            val_acc, val_loss = val_and_logging(args, epoch, default_model, augmenter, val_dataloader, test_dataloader, loss_func,
                                               float(np.mean(train_loss_list)), estimator=knn_estimator)
            torch.save(default_model.backbone.state_dict(), latest_weight)
            if val_loss < best_val_loss:
                best_val_loss = val_loss
                torch.save(default_model.backbone.state_dict(), best_weight)
            torch.save({"epoch": epoch, "best_val_loss": best_val_loss, "optimizer": optimizer.train_state()}, train_state)
        lr_scheduler.step(epoch)
    end = time_sync()
    logging.info(f"Total processing time: {(end - start): .3f} s  ({windows / max(end - start, 1e-9):.1f} windows/s)")
    return default_model
