"""Validation (reference: train_utils/eval_functions.py:10-137): under pretraining the mean pretraining loss plus accuracy /
macro-F1 / confusion matrix of the KNN estimator on the backbone features; under finetuning the classifier's own loss and metrics."""
import logging

import numpy as np
import torch
from sklearn.metrics import accuracy_score, confusion_matrix, f1_score

from train_utils.knn import extract_sample_features
from train_utils.loss_calc_utils import calc_pretrain_loss


def eval_task_metrics(args, labels, predictions):
    if args.task in {"distance_classification", "speed_classification"}:
        num_classes = args.dataset_config[args.task]["num_classes"]
        mean_acc = 1 - (np.abs(labels - predictions) / np.maximum(labels, (num_classes - 1) - labels))
        mean_acc = np.nan_to_num(mean_acc, nan=1.0).mean()
    else:
        mean_acc = accuracy_score(labels, predictions)
    mean_f1 = f1_score(labels, predictions, average="macro", zero_division=1)
    try:
        conf = confusion_matrix(labels, predictions)
    except Exception:  # noqa: BLE001
        conf = []
    return mean_acc, mean_f1, conf


def eval_supervised_model(args, classifier, augmenter, dataloader, loss_func):
    """Loss and task metrics of a classifier (reference :29-62)."""
    classifier.eval()
    losses, preds, labs = [], [], []
    with torch.no_grad():
        for time_loc_inputs, labels in dataloader:
            freq_loc_inputs, labels = augmenter.forward("no", time_loc_inputs, labels)
            logits = classifier(freq_loc_inputs)
            losses.append(loss_func(logits, labels).item())
            labels = labels.argmax(dim=1) if labels.dim() > 1 else labels
            preds.append(logits.argmax(dim=1).cpu().numpy())
            labs.append(labels.cpu().numpy())
    return float(np.mean(losses)), eval_task_metrics(args, np.concatenate(labs), np.concatenate(preds))


def eval_pretrained_model(args, default_model, estimator, augmenter, dataloader, loss_func):
    default_model.eval()
    feats, labels, losses = [], [], []
    with torch.no_grad():
        for time_loc_inputs, label in dataloader:
            label = label.argmax(dim=1) if label.dim() > 1 else label
            labels.append(label.cpu().numpy())
            losses.append(calc_pretrain_loss(args, default_model, augmenter, loss_func, time_loc_inputs).item())
            feats.append(extract_sample_features(args, default_model.backbone, augmenter.forward("no", time_loc_inputs)))
    predictions = estimator.predict(torch.cat(feats)).cpu().numpy()
    return float(np.mean(losses)), eval_task_metrics(args, np.concatenate(labels), predictions)


def val_and_logging(args, epoch, model, augmenter, val_loader, test_loader, loss_func, train_loss, estimator=None):
    if args.train_mode in {"contrastive"} and args.stage == "pretrain":
        logging.info(f"Train {args.train_mode} loss: {train_loss: .5f} \n")
    else:
        logging.info(f"Training loss: {train_loss: .5f} \n")
    if args.train_mode == "supervised" or args.stage == "finetune":
        val_loss, val_metrics = eval_supervised_model(args, model, augmenter, val_loader, loss_func)
        test_loss, test_metrics = eval_supervised_model(args, model, augmenter, test_loader, loss_func)
    else:
        val_loss, val_metrics = eval_pretrained_model(args, model, estimator, augmenter, val_loader, loss_func)
        test_loss, test_metrics = eval_pretrained_model(args, model, estimator, augmenter, test_loader, loss_func)
    logging.info(f"Val loss: {val_loss: .5f}")
    logging.info(f"Val acc: {val_metrics[0]: .5f}, val f1: {val_metrics[1]: .5f}")
    logging.info(f"Val confusion matrix:\n {val_metrics[2]} \n")
    logging.info(f"Test loss: {test_loss: .5f}")
    logging.info(f"Test acc: {test_metrics[0]: .5f}, test f1: {test_metrics[1]: .5f}")
    logging.info(f"Test confusion matrix:\n {test_metrics[2]} \n")
    return val_metrics[0], val_loss
