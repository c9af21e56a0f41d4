"""Downstream entry point with the reference's command line (code/run_downstream.py), simulated-data training branch:

    python run_downstream.py --ds-train --simu-exp --ds-trainmode finetune --ds-task TDOA --ds-nsimroom 8 --time <pretrain tag>

Per task and per (trial, batch size, learning rate) of the sweep: train / validate / test every epoch, early stopping on the
smoothed validation loss with one learning-rate decay (code/run_downstream.py:262-308), ensembling of the last five epochs up to
the best one, final test on the large test set, results to the same ``*-lr_bs_tri_result.mat`` file.  Scalars go to a JSONL log
(tensorboardX is absent here).  Same model code, kernels and learner as pretraining, at T = 1.04 s (nt = 64) for TDOA.
"""
import copy
import json
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import sarssl_boot  # noqa: E402,F401


def main(argv=None):
    from sar_ssl_amd.opt import opt_downstream
    opts = opt_downstream()
    args = opts.parse(argv)
    dirs = opts.dir()
    if "LOCAL_RANK" not in os.environ:
        os.environ["HIP_VISIBLE_DEVICES"] = ",".join(g for g in args.gpu_id.split(",") if g != "")

    import numpy as np
    import scipy.io
    import torch
    from sar_ssl_amd import dataset as at_dataset, learner as at_learner, model as at_model
    from sar_ssl_amd.common.utils import set_seed, set_random_seed, get_nparams

    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("run_downstream.py (sar_ssl_amd) needs an MI355X GPU: the HIP path has no CPU fallback")
    if not args.ds_train:
        raise SystemExit("only `--ds-train --simu-exp` is implemented on this path")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    set_seed(args.seed)
    assert args.source_state == "static", "Source state model unrecognized~"
    fs = args.acoustic_setting["fs"]
    seeds = {"train": int(args.seed + 2e8), "val": int(args.seed + 1e8), "test": int(args.seed + 1)}
    T = 1.04 if args.ds_task == ["TDOA"] else 4.112                                         # code/run_downstream.py:67-70
    selecting = at_dataset.Selecting(select_range=[0, int(T * fs)])
    win_len, nfft, win_shift_ratio, fre_used_ratio = 512, 512, 0.5, 1
    nf = nfft // 2
    nt = int((T * fs - win_len * (1 - win_shift_ratio)) / (win_len * win_shift_ratio))
    net = at_model.SARSSL(sig_shape=(nf, nt, 2, 2), pretrain=False, device=device, downstream_token=args.ds_token,
                          downstream_head=args.ds_head, downstream_embed=args.ds_embed, downstream_dlabel=1)
    nparam, nparam_sum = get_nparams(net, param_key_list=["spec_encoder", "spat_encoder", "mlp_head"])
    print("duration:", T, "s; nt, nf:", nt, nf, "; # Parameters (M):", round(nparam_sum, 2))

    log_dir = "log_task_" + args.ds_trainmode
    init_state_dict = copy.deepcopy(net.state_dict())
    for task in args.ds_task:
        set_seed(args.seed)
        task_time_dir = dirs["log_task"].replace("TASK", task)
        st = args.ds_setting[task]
        nepoch, num, bs_set, lr_set, ntrials = st["nepoch"], st["num"], st["bs_set"], st["lr_set"], st["ntrial"]
        neval = args.ds_eval_num
        data_num = {"train": num, "val": neval or 1000, "test": neval or 1000, "test_large": neval or 4000}
        stages = ["train", "val", "test", "test_large"]
        test_bs, early_stop_patience, smooth_alpha, nepoch_ensemble, num_stop_th = 16, 10, 0.6, 5, 1
        nlrs, nbss = len(lr_set), len(bs_set)
        shape = (nlrs, nbss, ntrials)
        val_losses, test_losses, val_metrics, test_metrics = (np.zeros(shape) for _ in range(4))
        ensemble_epochs = np.zeros(shape + (2,))
        os.makedirs(task_time_dir, exist_ok=True)
        task_dir = None
        for trial_idx in range(ntrials):
            for bs_idx in range(nbss):
                for lr_idx in range(nlrs):
                    set_seed(args.seed)
                    lr_init, bs = lr_set[lr_idx], bs_set[bs_idx]
                    print(task, ": nepoch=", nepoch, "num=", num, "lr=", lr_init, "bs=", bs, "trial_idx=", trial_idx, "ntrial=", ntrials)
                    task_dir = (dirs[log_dir].replace("TASK", task).replace("NUM", str(num)).replace("LR", str(lr_init))
                                .replace("BAS", str(bs)).replace("TRI", str(trial_idx)))
                    os.makedirs(task_dir, exist_ok=True)
                    datasets = {}
                    for stage in stages:
                        key = "micsig_" + stage.split("_")[0] + "_simu"
                        data_dir = dirs[key][trial_idx] if stage == "train" else dirs[key]
                        datasets[stage] = at_dataset.FixMicSigDataset(data_dir=data_dir, load_anno=True, load_dp=False, fs=fs,
                                                                      dataset_sz=data_num[stage], transforms=[selecting])
                    kwargs = {"num_workers": args.workers, "pin_memory": True}
                    dl_train = torch.utils.data.DataLoader(datasets["train"], batch_size=bs, shuffle=True, **kwargs)
                    dl_val = torch.utils.data.DataLoader(datasets["val"], batch_size=test_bs, shuffle=False, **kwargs)
                    dl_test = torch.utils.data.DataLoader(datasets["test"], batch_size=test_bs, shuffle=False, **kwargs)
                    dl_test_large = torch.utils.data.DataLoader(datasets["test_large"], batch_size=test_bs, shuffle=False, **kwargs)

                    net.load_state_dict(init_state_dict)
                    for p in net.parameters():
                        p.requires_grad = True
                    learner = at_learner.STFTLearner(net, win_len=win_len, win_shift_ratio=win_shift_ratio, nfft=nfft,
                                                     fre_used_ratio=fre_used_ratio, fs=fs, task=task, ch_mode="M")
                    learner.cuda()
                    if args.use_amp:
                        learner.amp()
                    if args.checkpoint_start:
                        learner.resume_checkpoint(checkpoints_dir=task_dir, from_latest=True, as_all_state=True)
                    elif args.ds_trainmode == "finetune":
                        learner.load_checkpoint_best(checkpoints_dir=dirs["log_pretrain"], as_all_state=False, param_frozen=False)
                    elif args.ds_trainmode == "lineareval":
                        learner.load_checkpoint_best(checkpoints_dir=dirs["log_pretrain"], as_all_state=False, param_frozen=True)

                    log = open(os.path.join(task_dir, "scalars.jsonl"), "a")
                    loss_val_list, lr, cnt_stop, best_epoch, epoch = [], lr_init * 1, 0, learner.start_epoch, learner.start_epoch
                    for epoch in range(learner.start_epoch, nepoch + 1):
                        set_random_seed(seeds["train"])
                        loss_train, metric_train = learner.train_epoch(dl_train, lr=lr, epoch=epoch, return_metric=True)
                        set_random_seed(seeds["val"])
                        loss_val, metric_val = learner.test_epoch(dl_val, return_metric=True)
                        set_random_seed(seeds["test"])
                        loss_test, metric_test = learner.test_epoch(dl_test, return_metric=True)
                        loss_val_list += [loss_val]
                        smooth = learner.smooth_data(data_list=loss_val_list, alpha=smooth_alpha)
                        stop_flag, is_best = learner.early_stopping(current_score=smooth[-1] * (-1), patience=early_stop_patience)
                        learner.save_checkpoint(epoch=epoch, checkpoints_dir=task_dir, is_best_epoch=is_best, save_extra_hist=True)
                        if is_best:
                            best_epoch = copy.deepcopy(epoch)
                        rec = {"epoch": epoch, "lr": lr, "loss_train": loss_train, "metric_train": float(metric_train),
                               "loss_val": loss_val, "metric_val": float(metric_val), "loss_val_smooth": smooth[-1],
                               "loss_test": loss_test, "metric_test": float(metric_test)}
                        print(json.dumps(rec), flush=True)
                        log.write(json.dumps(rec) + "\n"); log.flush()
                        if stop_flag:
                            cnt_stop += 1
                            if cnt_stop <= num_stop_th:
                                lr = lr / 10
                                print("lr decaing")
                                learner.early_stop_counter = 0
                            else:
                                break
                    print("\nTraining finished\n")
                    st_epoch = int(np.maximum(1, best_epoch - nepoch_ensemble + 1))
                    ed_epoch = copy.deepcopy(best_epoch)
                    learner.ensembling(checkpoints_dir=task_dir, epochs=[i for i in range(st_epoch, ed_epoch + 1)])
                    set_random_seed(seeds["test"])
                    best_loss_test, best_metric_test = learner.test_epoch(dl_test_large, return_metric=True)
                    set_random_seed(seeds["val"])
                    best_loss_val, best_metric_val = learner.test_epoch(dl_val, return_metric=True)
                    print("{} estimation, Test loss: {:.4f}, Test metric: {:.4f}".format(task, best_loss_test, float(best_metric_test)))
                    print("{} estimation, Val loss: {:.4f}, Val metric: {:.4f}".format(task, best_loss_val, float(best_metric_val)))
                    learner.remove_checkpoint_epochs(checkpoints_dir=task_dir, epochs=[i for i in range(1, st_epoch)] +
                                                     [i for i in range(best_epoch + 1, epoch + 1)])
                    val_losses[lr_idx, bs_idx, trial_idx], val_metrics[lr_idx, bs_idx, trial_idx] = best_loss_val, float(best_metric_val)
                    test_losses[lr_idx, bs_idx, trial_idx], test_metrics[lr_idx, bs_idx, trial_idx] = best_loss_test, float(best_metric_test)
                    ensemble_epochs[lr_idx, bs_idx, trial_idx, :] = [st_epoch, ed_epoch]
                    log.close()
        metric = np.mean(val_metrics, axis=-1)
        idxes = metric.argmin()
        best_lr_idx, best_bs_idx = idxes // metric.shape[1], idxes % metric.shape[1]
        print("\n{} estimation, BS: {}, LR: {}, best val MAE: {:.4f}, best test MAE: {:.4f}\n".format(
            task, bs_set[best_bs_idx], lr_set[best_lr_idx], metric[best_lr_idx, best_bs_idx],
            np.mean(test_metrics, axis=-1)[best_lr_idx, best_bs_idx]))
        atts = task_dir.replace(task_time_dir, "").split("-")
        result_name = "-".join([atts[0], atts[1], atts[2], atts[3], atts[-2], atts[-1]]) + "-lr_bs_tri_result.mat"
        scipy.io.savemat(task_time_dir + "/" + result_name.lstrip("/"), {
            "val_losses": val_losses, "val_metrics": val_metrics, "test_losses": test_losses, "test_metrics": test_metrics,
            "lr_set": lr_set, "bs_set": bs_set, "ntrial": ntrials, "best_lr_idx": best_lr_idx, "best_bs_idx": best_bs_idx,
            "ensemble_epoch": ensemble_epochs})


if __name__ == "__main__":
    main()
