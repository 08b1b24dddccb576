"""Supervised training of a backbone from scratch (reference: train_utils/supervised_train.py:18-108; `train.py -learn_framework=no`):
every parameter goes to the optimizer, batches pass through the `fixed` augmentation pipeline (Mixup in the time domain, phase
shift after the transform -- data_augmenter/Augmenter.py), the loss is cross-entropy on `backbone(freq_x)` logits, validation every 5
epochs keeps the latest / best weights.  On the HIP path the backward runs through the classifier head (focal_amd/head_engine.py)
into the same encoder kernels pretraining uses; the patch embedding, frozen in pretraining, is trained here."""
import logging
import os

import numpy as np
import torch

from general_utils.time_utils import time_sync
from train_utils.eval_functions import val_and_logging
from train_utils.lr_scheduler import define_lr_scheduler
from train_utils.optimizer import define_optimizer


def supervised_train(args, classifier, augmenter, train_dataloader, val_dataloader, test_dataloader, loss_func, num_batches):
    classifier_config = args.dataset_config[args.model]
    optimizer = define_optimizer(args, classifier.parameters())
    lr_scheduler = define_lr_scheduler(args, optimizer)
    logging.info("---------------------------Start Pretraining Classifier-------------------------------")
    start = time_sync()
    best_val_acc = 0
    best_weight = os.path.join(args.weight_folder, f"{args.dataset}_{args.model}_{args.task}_best.pt")
    latest_weight = os.path.join(args.weight_folder, f"{args.dataset}_{args.model}_{args.task}_latest.pt")
    val_epochs = 5
    epochs = getattr(args, "epochs", None) or classifier_config["lr_scheduler"]["train_epochs"]
    for epoch in range(epochs):
        if epoch > 0:
            logging.info("-" * 40 + f"Epoch {epoch}" + "-" * 40)
        classifier.train()
        args.epoch = epoch
        train_loss_list = []
        for i, (time_loc_inputs, labels) in enumerate(train_dataloader):
            aug_freq_loc_inputs, labels = augmenter.forward("fixed", time_loc_inputs, labels)
            optimizer.zero_grad()  # (the reference zeroes between forward and backward; gradients accumulate in the arena only in backward)
            logits = classifier(aug_freq_loc_inputs)
            loss = loss_func(logits, labels)
            loss.backward()
            optimizer.step()
            train_loss_list.append(loss.item())
        if epoch % val_epochs == 0:
            val_metric, val_loss = val_and_logging(args, epoch, classifier, augmenter, val_dataloader, test_dataloader, loss_func,
                                                  float(np.mean(train_loss_list)))
            torch.save(classifier.state_dict(), latest_weight)
            if val_metric > best_val_acc:
                best_val_acc = val_metric
                torch.save(classifier.state_dict(), best_weight)
        lr_scheduler.step(epoch)
    end = time_sync()
    logging.info("------------------------------------------------------------------------")
    logging.info(f"Total processing time: {(end - start): .3f} s")
    return classifier
