"""Finetuning of a pretrained backbone (reference: train_utils/finetune.py:18-96): load the pretraining weights, train only the
class layer and the modality fusion layer with cross-entropy, validate every 5 epochs, keep latest / best weights."""
import logging
import os

import numpy as np
import torch

from general_utils.time_utils import time_sync
from general_utils.weight_utils import load_model_weight, set_learnable_params_finetune
from train_utils.eval_functions import val_and_logging
from train_utils.lr_scheduler import define_lr_scheduler
from train_utils.optimizer import define_optimizer


def finetune(args, classifier, augmenter, train_dataloader, val_dataloader, test_dataloader, classifier_loss_func, num_batches):
    pretrain_weight = os.path.join(args.weight_folder, f"{args.dataset}_{args.model}_pretrain_latest.pt")
    classifier = load_model_weight(args, classifier, pretrain_weight, load_class_layer=False)
    learnable_parameters = set_learnable_params_finetune(args, classifier)
    optimizer = define_optimizer(args, learnable_parameters)
    lr_scheduler = define_lr_scheduler(args, optimizer)
    tag = f"{args.dataset}_{args.model}_{args.task}_{args.label_ratio}_finetune"
    best_weight = os.path.join(args.weight_folder, f"{tag}_best.pt")
    latest_weight = os.path.join(args.weight_folder, f"{tag}_latest.pt")

    logging.info("---------------------------Start Fine Tuning-------------------------------")
    start = time_sync()
    best_val_acc, val_epochs = 0, 5
    epochs = getattr(args, "epochs", None) or args.dataset_config[args.learn_framework]["finetune_lr_scheduler"]["train_epochs"]
    for epoch in range(epochs):
        classifier.train()
        train_loss_list = []
        for i, (time_loc_inputs, labels) in enumerate(train_dataloader):
            aug_freq_loc_inputs, labels = augmenter.forward("no", time_loc_inputs, labels)
            logits = classifier(aug_freq_loc_inputs)
            loss = classifier_loss_func(logits, labels)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            train_loss_list.append(loss.item())
        if epoch % val_epochs == 0:
            val_metric, val_loss = val_and_logging(args, epoch, classifier, augmenter, val_dataloader, test_dataloader,
                                                  classifier_loss_func, float(np.mean(train_loss_list)))
            torch.save(classifier.state_dict(), latest_weight)
            if val_metric > best_val_acc:
                best_val_acc = val_metric
                torch.save(classifier.state_dict(), best_weight)
        lr_scheduler.step(epoch)
    end = time_sync()
    logging.info("------------------------------------------------------------------------")
    logging.info(f"Total processing time: {(end - start): .3f} s")
    return classifier
