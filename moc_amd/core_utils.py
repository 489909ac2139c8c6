"""The CLAM-lineage trainer hooks the baselines are driven by, under the reference's names and
signatures (reference utils/core_utils.py): Accuracy_Logger,
EarlyStopping, train_loop, validate, summary and a `train` for the max-instance MIL models.
A model is anything whose forward returns the 5-tuple (logits, Y_prob, Y_hat, _, _) -- moc_amd.model_mil
on the HIP path.  Loaders yield (data, label) with batch size 1.  CLAM / ViLa variants, tensorboard and
the SVM loss are outside the MOC path and not reproduced."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
from sklearn.metrics import auc as calc_auc
from sklearn.metrics import roc_auc_score, roc_curve
from sklearn.preprocessing import label_binarize


def _device():
    assert torch.cuda.is_available(), "moc_amd.core_utils drives the HIP path: a GPU is required"
    return torch.device("cuda")


def calculate_error(Y_hat, Y):
    """utils/utils.py:379-381."""
    return 1.0 - Y_hat.float().eq(Y.float()).float().mean().item()


class Accuracy_Logger:
    """Per-class hit counter (reference :16-50)."""

    def __init__(self, n_classes):
        self.n_classes = n_classes
        self.initialize()

    def initialize(self):
        self.data = [{"count": 0, "correct": 0} for _ in range(self.n_classes)]

    def log(self, Y_hat, Y):
        Y_hat, Y = int(Y_hat), int(Y)
        self.data[Y]["count"] += 1
        self.data[Y]["correct"] += (Y_hat == Y)

    def log_batch(self, Y_hat, Y):
        Y_hat, Y = np.array(Y_hat).astype(int), np.array(Y).astype(int)
        for c in np.unique(Y):
            sel = Y == c
            self.data[c]["count"] += sel.sum()
            self.data[c]["correct"] += (Y_hat[sel] == Y[sel]).sum()

    def get_summary(self, c):
        count, correct = self.data[c]["count"], self.data[c]["correct"]
        return (None if count == 0 else float(correct) / count), correct, count


class EarlyStopping:
    """Checkpoint on improvement of -val_loss (or of `criteria`), stop after `patience` epochs without
    one once past `stop_epoch` (reference :53-102, including its first call, which always saves)."""

    def __init__(self, patience=20, stop_epoch=50, verbose=False):
        self.patience, self.stop_epoch, self.verbose = patience, stop_epoch, verbose
        self.counter, self.best_score, self.early_stop = 0, None, False
        self.val_loss_min = np.inf

    def __call__(self, epoch, val_loss, model, ckpt_name="checkpoint.pt", criteria=None):
        score = criteria if criteria else -val_loss          # criteria == 0 falls back to the loss (`not criteria`)
        first = self.best_score is None
        if first:
            self.best_score = -1                              # what the first verbose message reports
        if first or score > self.best_score:
            self.save_checkpoint(val_loss, model, ckpt_name, criteria=criteria)
            self.best_score = score
            if not first:
                self.counter = 0
            return
        self.counter += 1
        print(f"EarlyStopping counter: {self.counter} out of {self.patience}")
        if self.counter >= self.patience and epoch > self.stop_epoch:
            self.early_stop = True

    def save_checkpoint(self, val_loss, model, ckpt_name, criteria=None):
        if self.verbose:
            if criteria:
                print(f"Validation criteria increased ({self.best_score:.6f} --> {criteria:.6f}).  Saving model ...")
            else:
                print(f"Validation loss decreased ({self.val_loss_min:.6f} --> {val_loss:.6f}).  Saving model ...")
        torch.save(model.state_dict(), ckpt_name)
        self.val_loss_min = val_loss


def _to_device(data, device):
    if type(data) == tuple:
        return tuple(d.to(device) for d in data)
    return data.to(device)


def _class_report(acc_logger, n_classes, writer=None, tag=None, epoch=0):
    accs = []
    for i in range(n_classes):
        acc, correct, count = acc_logger.get_summary(i)
        print("class {}: acc {}, correct {}/{}".format(i, acc, correct, count))
        accs.append(acc)
        if writer and acc is not None and tag:
            writer.add_scalar(tag.format(i), acc, epoch)
    return accs


def train_loop(epoch, model, loader, optimizer, n_classes, writer=None, loss_fn=None, bag_size=None):
    """One pass, one optimizer step per bag (reference :372-432)."""
    device = _device()
    model.train()
    acc_logger = Accuracy_Logger(n_classes=n_classes)
    train_loss = train_error = 0.0
    print("\n")
    for batch_idx, (data, label) in enumerate(loader):
        data, label = _to_device(data, device), label.to(device)
        logits, Y_prob, Y_hat, _, _ = model(data)
        if Y_hat.size(0) > 1:
            acc_logger.log_batch(Y_hat.cpu(), label.cpu())
        else:
            acc_logger.log(Y_hat, label)
        loss = loss_fn(logits, label)
        train_loss += loss.item()
        if (batch_idx + 1) % 20 == 0:
            bag = data[0] if type(data) == tuple else data
            print("batch {}, loss: {:.4f}, bag_size: {}".format(batch_idx, loss.item(), bag.size(-2)))
        train_error += calculate_error(Y_hat, label)
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
    train_loss /= len(loader)
    train_error /= len(loader)
    print("Epoch: {}, train_loss: {:.4f}, train_error: {:.4f}".format(epoch, train_loss, train_error))
    _class_report(acc_logger, n_classes, writer, "train/class_{}_acc", epoch)
    if writer:
        writer.add_scalar("train/loss", train_loss, epoch)
        writer.add_scalar("train/error", train_error, epoch)


def train_loop_clam(epoch, model, loader, optimizer, n_classes, bag_weight, writer=None, loss_fn=None, bag_size=None):
    """One pass of CLAM training (reference :294-370): the bag loss and the instance-level clustering loss of
    `model(data, label=label, instance_eval=True)` mixed as bag_weight * loss + (1 - bag_weight) * instance_loss."""
    device = _device()
    model.train()
    acc_logger = Accuracy_Logger(n_classes=n_classes)
    inst_logger = Accuracy_Logger(n_classes=n_classes)
    train_loss = train_error = train_inst_loss = 0.0
    inst_count = 0
    for batch_idx, (data, label) in enumerate(loader):
        data, label = data.to(device), label.to(device)
        logits, Y_prob, Y_hat, _, instance_dict = model(data, label=label, instance_eval=True)
        if Y_hat.size(0) > 1:
            acc_logger.log_batch(Y_hat.cpu(), label.cpu())
        else:
            acc_logger.log(Y_hat, label)
        loss = loss_fn(logits, label)
        loss_value = loss.item()
        instance_loss = instance_dict["instance_loss"]
        inst_count += 1
        instance_loss_value = instance_loss.item()
        train_inst_loss += instance_loss_value
        total_loss = bag_weight * loss + (1 - bag_weight) * instance_loss
        inst_logger.log_batch(instance_dict["inst_preds"], instance_dict["inst_labels"])
        train_loss += loss_value
        if (batch_idx + 1) % 20 == 0:
            print("batch {}, loss: {:.4f}, instance_loss: {:.4f}, weighted_loss: {:.4f}, ".format(
                batch_idx, loss_value, instance_loss_value, total_loss.item()))
        train_error += calculate_error(Y_hat, label)
        total_loss.backward()
        optimizer.step()
        optimizer.zero_grad()
    train_loss /= len(loader)
    train_error /= len(loader)
    if inst_count > 0:
        train_inst_loss /= inst_count
        for i in range(2):
            acc, correct, count = inst_logger.get_summary(i)
            print("class {} clustering acc {}: correct {}/{}".format(i, acc, correct, count))
    print("Epoch: {}, train_loss: {:.4f}, train_clustering_loss:  {:.4f}, train_error: {:.4f}".format(
        epoch, train_loss, train_inst_loss, train_error))
    for i in range(n_classes):
        acc, correct, count = acc_logger.get_summary(i)
        print("train class {}: acc {}, correct {}/{}".format(i, acc, correct, count))
        if writer and acc is not None:
            writer.add_scalar("train/class_{}_acc".format(i), acc, epoch)
    if writer:
        writer.add_scalar("train/loss", train_loss, epoch)
        writer.add_scalar("train/error", train_error, epoch)
        writer.add_scalar("train/clustering_loss", train_inst_loss, epoch)


def _epoch_auc(labels, prob, n_classes):
    """The AUC block validate / validate_clam / summary share (reference :505-519, :601-615)."""
    if n_classes == 2:
        return roc_auc_score(labels, prob[:, 1])
    onehot = label_binarize(labels, classes=list(range(n_classes)))
    per_class = []
    for c in range(n_classes):
        if c in labels:
            fpr, tpr, _ = roc_curve(onehot[:, c], prob[:, c])
            per_class.append(calc_auc(fpr, tpr))
        else:
            per_class.append(float("nan"))
    return np.nanmean(np.array(per_class))


def validate_clam(cur, epoch, model, loader, n_classes, early_stopping=None, writer=None, loss_fn=None,
                  results_dir=None, disableAUC=False):
    """Validation pass of CLAM with instance evaluation (reference :558-656) -> True when early stopping fires."""
    device = _device()
    model.eval()
    acc_logger = Accuracy_Logger(n_classes=n_classes)
    inst_logger = Accuracy_Logger(n_classes=n_classes)
    val_loss = val_error = val_inst_loss = 0.0
    inst_count = 0
    prob = np.zeros((len(loader), n_classes))
    labels = np.zeros(len(loader))
    with torch.no_grad():
        for batch_idx, (data, label) in enumerate(loader):
            data, label = data.to(device), label.to(device)
            logits, Y_prob, Y_hat, _, instance_dict = model(data, label=label, instance_eval=True)
            acc_logger.log(Y_hat, label)
            val_loss += loss_fn(logits, label).item()
            inst_count += 1
            val_inst_loss += instance_dict["instance_loss"].item()
            inst_logger.log_batch(instance_dict["inst_preds"], instance_dict["inst_labels"])
            prob[batch_idx] = Y_prob.cpu().numpy()
            labels[batch_idx] = label.item()
            val_error += calculate_error(Y_hat, label)
    val_error /= len(loader)
    val_loss /= len(loader)
    auc = 0 if disableAUC else _epoch_auc(labels, prob, n_classes)
    if disableAUC:
        bacc = 0
    else:
        print("\nValidation Set")
        acc_list = []
        for i in range(n_classes):
            acc, correct, count = acc_logger.get_summary(i)
            print("class {}: acc {}, correct {}/{}".format(i, acc, correct, count))
            acc_list.append(acc)
            if writer and acc is not None:
                writer.add_scalar("val/class_{}_acc".format(i), acc, epoch)
        bacc = np.mean(acc_list)
        print("balanced accuracy: ", bacc)
    print("Val Set, val_loss: {:.4f}, val_error: {:.4f}, auc: {:.4f}, bacc: {:.4f}".format(val_loss, val_error, auc, bacc))
    if inst_count > 0:
        val_inst_loss /= inst_count
        for i in range(2):
            acc, correct, count = inst_logger.get_summary(i)
            print("class {} clustering acc {}: correct {}/{}".format(i, acc, correct, count))
    if writer:
        writer.add_scalar("val/loss", val_loss, epoch)
        writer.add_scalar("val/auc", auc, epoch)
        writer.add_scalar("val/error", val_error, epoch)
        writer.add_scalar("val/inst_loss", val_inst_loss, epoch)
    if early_stopping:
        assert results_dir
        ckpt = os.path.join(results_dir, "s_{}_checkpoint.pt".format(cur))
        if disableAUC:
            early_stopping(epoch, val_loss, model, ckpt_name=ckpt)
        else:
            early_stopping(epoch, val_loss, model, ckpt_name=ckpt, criteria=auc)
        if early_stopping.early_stop:
            print("Early stopping")
            return True
    return False


def validate(cur, epoch, model, loader, n_classes, early_stopping=None, writer=None, loss_fn=None, results_dir=None,
             disableAUC=False):
    """Validation pass; returns True when early stopping fires (reference :480-556)."""
    device = _device()
    model.eval()
    acc_logger = Accuracy_Logger(n_classes=n_classes)
    val_loss = val_error = 0.0
    prob, labels = np.zeros((len(loader), n_classes)), np.zeros(len(loader))
    with torch.no_grad():
        for batch_idx, (data, label) in enumerate(loader):
            data, label = _to_device(data, device), label.to(device)
            logits, Y_prob, Y_hat, _, _ = model(data)
            acc_logger.log(Y_hat, label)
            val_loss += loss_fn(logits, label).item()
            prob[batch_idx], labels[batch_idx] = Y_prob.cpu().numpy(), label.item()
            val_error += calculate_error(Y_hat, label)
    val_error /= len(loader)
    val_loss /= len(loader)
    if disableAUC:
        auc = bacc = 0
    else:
        auc = roc_auc_score(labels, prob[:, 1]) if n_classes == 2 else roc_auc_score(labels, prob, multi_class="ovr")
    if writer:
        writer.add_scalar("val/loss", val_loss, epoch)
        writer.add_scalar("val/auc", auc, epoch)
        writer.add_scalar("val/error", val_error, epoch)
    if not disableAUC:
        print("\nValidation Set")
        bacc = np.mean(_class_report(acc_logger, n_classes))
        print("balanced accuracy: ", bacc)
    print("Val Set, val_loss: {:.4f}, val_error: {:.4f}, auc: {:.4f}, bacc: {:.4f}".format(val_loss, val_error, auc, bacc))
    if early_stopping:
        assert results_dir
        ckpt = os.path.join(results_dir, "s_{}_checkpoint.pt".format(cur))
        if disableAUC:
            early_stopping(epoch, val_loss, model, ckpt_name=ckpt)
        else:
            early_stopping(epoch, val_loss, model, ckpt_name=ckpt, criteria=auc)
        if early_stopping.early_stop:
            print("Early stopping")
            return True
    return False


def summary(model, loader, n_classes, require_patient_results=True):
    """-> (patient_results, test_error, auc, acc_logger) (reference :734-788; multi-class AUC is the
    mean of the per-class one-vs-rest curves over the classes present)."""
    device = _device()
    acc_logger = Accuracy_Logger(n_classes=n_classes)
    model.eval()
    test_error = 0.0
    all_probs, all_labels = np.zeros((len(loader), n_classes)), np.zeros(len(loader))
    slide_ids = loader.dataset.slide_data["slide_id"] if require_patient_results else None
    patient_results = {}
    for batch_idx, (data, label) in enumerate(loader):
        data, label = _to_device(data, device), label.to(device)
        with torch.no_grad():
            logits, Y_prob, Y_hat, _, _ = model(data)
        acc_logger.log(Y_hat, label)
        probs = Y_prob.cpu().numpy()
        all_probs[batch_idx], all_labels[batch_idx] = probs, label.item()
        if require_patient_results:
            sid = slide_ids.iloc[batch_idx]
            patient_results[sid] = {"slide_id": np.array(sid), "prob": probs, "label": label.item()}
        test_error += calculate_error(Y_hat, label)
    test_error /= len(loader)
    if n_classes == 2:
        auc = roc_auc_score(all_labels, all_probs[:, 1])
    else:
        onehot = label_binarize(all_labels, classes=list(range(n_classes)))
        per_class = []
        for c in range(n_classes):
            if c in all_labels:
                fpr, tpr, _ = roc_curve(onehot[:, c], all_probs[:, c])
                per_class.append(calc_auc(fpr, tpr))
            else:
                per_class.append(float("nan"))
        auc = np.nanmean(np.array(per_class))
    return patient_results, test_error, auc, acc_logger


def get_optim(model, args):
    """utils/utils.py:270-279."""
    params = filter(lambda p: p.requires_grad, model.parameters())
    if args.opt == "adam":
        return torch.optim.Adam(params, lr=args.lr, weight_decay=args.reg)
    if args.opt == "sgd":
        return torch.optim.SGD(params, lr=args.lr, momentum=0.9, weight_decay=args.reg)
    if args.opt == "adamW":
        return torch.optim.AdamW(params, lr=args.lr, weight_decay=args.reg, betas=(0.9, 0.999))
    raise NotImplementedError


def train(datasets, cur, args, pseudo=False, notsavesplit=False, require_patient_results=True, disableAUC=False):
    """One fold (reference :105-291) for the model types on this path: 'mil' (MIL_fc / MIL_fc_mc), 'clam_sb',
    'clam_mb', 'abmil' (CLAM_SB without instance loss) and 'transmil'.  `datasets` = (train_loader, val_loader, test_loader) of
    (data, label) bags.  As in the reference: CrossEntropy bag loss, get_optim, CosineAnnealingLR(optimizer, 20)
    stepped once per epoch, EarlyStopping(patience=20, stop_epoch=40) when args.early_stopping, the CLAM loops unless
    args.no_inst_cluster, checkpoint s_{cur}_checkpoint.pt.  Returns (results_dict, test_auc, val_auc,
    1 - test_error, 1 - val_error).  Not here: the svm losses, tensorboard, ViLa / CHIEF / TITAN."""
    from .model_clam import CLAM_MB, CLAM_SB
    from .model_mil import MIL_fc, MIL_fc_mc, TransMIL
    model_type = getattr(args, "model_type", "mil")
    assert model_type in ("mil", "clam_sb", "clam_mb", "abmil", "transmil"), \
        f"model_type {model_type!r} is not on this path (mil, clam_sb, clam_mb, abmil, transmil)"
    assert getattr(args, "bag_loss", "ce") == "ce" and getattr(args, "inst_loss", None) in (None, "ce"), \
        "the smooth-SVM losses need the third-party `topk` package"
    os.makedirs(args.results_dir, exist_ok=True)
    train_loader, val_loader, test_loader = datasets
    loss_fn = nn.CrossEntropyLoss()
    kw = dict(dropout=getattr(args, "drop_out", False), n_classes=args.n_classes)
    if getattr(args, "model_size", None) is not None and model_type != "mil":
        kw["size_arg"] = args.model_size
    if model_type in ("clam_sb", "clam_mb"):
        if getattr(args, "subtyping", False):
            kw["subtyping"] = True
        if getattr(args, "B", 0) > 0:
            kw["k_sample"] = args.B
        model = (CLAM_SB if model_type == "clam_sb" else CLAM_MB)(**kw, instance_loss_fn=nn.CrossEntropyLoss())
    elif model_type == "abmil":
        model = CLAM_SB(**kw, instance_loss_fn=None)
    elif model_type == "transmil":
        model = TransMIL(**kw)
    elif args.n_classes > 2:
        model = MIL_fc_mc(**kw)
    else:
        kw["top_k"] = getattr(args, "topk", 1)
        model = MIL_fc(**kw)
    if hasattr(model, "relocate"):
        model.relocate()
    else:
        model = model.to(_device())
    optimizer = get_optim(model, args)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, 20)
    stopper = EarlyStopping(patience=20, stop_epoch=40, verbose=True) if getattr(args, "early_stopping", False) else None
    clam_loops = model_type in ("clam_sb", "clam_mb") and not getattr(args, "no_inst_cluster", False)
    for epoch in range(args.max_epochs):
        print("\n\nCurrent Epoch {}".format(epoch))
        print(f"lr: {scheduler.get_last_lr()}")
        if clam_loops:
            train_loop_clam(epoch, model, train_loader, optimizer, args.n_classes, args.bag_weight, None, loss_fn,
                            getattr(args, "bag_size", None))
            scheduler.step()
            stop = validate_clam(cur, epoch, model, val_loader, args.n_classes, stopper, None, loss_fn, args.results_dir,
                                 disableAUC=disableAUC)
        else:
            train_loop(epoch, model, train_loader, optimizer, args.n_classes, None, loss_fn, getattr(args, "bag_size", None))
            scheduler.step()
            stop = validate(cur, epoch, model, val_loader, args.n_classes, stopper, None, loss_fn, args.results_dir,
                            disableAUC=disableAUC)
        if stop:
            break
    ckpt = os.path.join(args.results_dir, "s_{}_checkpoint.pt".format(cur))
    if stopper:
        model.load_state_dict(torch.load(ckpt))
    else:
        torch.save(model.state_dict(), ckpt)
    if disableAUC:
        return {}, 0, 0, 0, 0
    _, val_error, val_auc, _ = summary(model, val_loader, args.n_classes, require_patient_results=require_patient_results)
    print("Val error: {:.4f}, ROC AUC: {:.4f}".format(val_error, val_auc))
    results, test_error, test_auc, acc_logger = summary(model, test_loader, args.n_classes,
                                                        require_patient_results=require_patient_results)
    print("Test error: {:.4f}, ROC AUC: {:.4f}".format(test_error, test_auc))
    print("Test balanced accuracy: ", np.mean(_class_report(acc_logger, args.n_classes)))
    return results, test_auc, val_auc, 1 - test_error, 1 - val_error
