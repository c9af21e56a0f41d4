"""Command-line options and directory layouts of the entry points: same flag names / defaults as ``opt_pretrain``
(code/opt.py:6-115) and ``opt_downstream`` (code/opt.py:117-320, simulated-data branch) in the reference."""
import argparse
import os
import time

import numpy as np


class opt_pretrain():
    def __init__(self):
        self.time = time.strftime("%m%d%H%M", time.localtime(time.time()))
        self.work_dir = os.path.abspath(os.path.expanduser(r"~"))
        self.work_dir_local = self.work_dir
        self.acoustic_setting = {"sound_speed": 343.0, "fs": 16000, "T": 4.112, "nmic": 2, "mic_dist_range": [0.03, 0.20]}

    def parse(self, argv=None):
        p = argparse.ArgumentParser(description="Self-supervised learing for multi-channel audio processing")
        p.add_argument("--gpu-id", type=str, default="7", metavar="GPU", help="GPU ID (default: 7)")
        p.add_argument("--workers", type=int, default=8, metavar="Worker", help="number of workers (default: 8)")
        p.add_argument("--bs", type=int, nargs="+", default=[128, 128, 128], metavar="TrainValTestBatch",
                       help="batch size for training, validation and test (default: [128, 128, 128])")
        p.add_argument("--no-cuda", action="store_true", default=False, help="disables CUDA training (default: False)")
        p.add_argument("--use-amp", action="store_true", default=False, help="Use mixed precision (MI355X: the 'hybrid' mode - fp16 stem, f32 residual stream; SARSSL_AMP_DTYPE=fp16|bf16 selects the others)")
        p.add_argument("--seed", type=int, default=1, metavar="Seed", help="random seed (default: 1)")
        p.add_argument("--checkpoint-start", action="store_true", default=False, help="train model from saved latest checkpoints")
        p.add_argument("--checkpoint-from-best-epoch", action="store_true", default=False, help="train model from saved best checkpoints")
        p.add_argument("--time", type=str, default=self.time, metavar="Time", help="time flag")
        p.add_argument("--work-dir", type=str, default=self.work_dir, metavar="WorkDir", help="work directory")
        p.add_argument("--sources", type=int, nargs="+", default=[1], metavar="Sources", help="number of sources (default: 1)")
        p.add_argument("--source-state", type=str, default="static", metavar="SourceState", help="state of sources")
        p.add_argument("--simu-exp", action="store_true", default=False, help="Experiments on simulated data")
        p.add_argument("--pretrain", action="store_true", default=False, help="change to pretrain stage")
        p.add_argument("--pretrain-frozen-encoder", action="store_true", default=False, help="(not implemented on this path)")
        p.add_argument("--nepoch", type=int, default=30, metavar="Epoch", help="number of epochs to train (default: 30)")
        p.add_argument("--lr", type=float, default=0.001, metavar="LR", help="learning rate (default:0.001)")
        p.add_argument("--test", action="store_true", default=False, help="change to test stage")
        p.add_argument("--test-mode", type=str, default="all", metavar="TestMode", help="test mode (default: all)")
        args = p.parse_args(argv)
        assert (args.pretrain + args.pretrain_frozen_encoder + args.test) == 1, "Pretraining stage (pretrain or test) is undefined"
        assert args.test_mode in ["all", "ins"], "Test mode is undefined"
        self.time, self.work_dir = args.time, os.path.abspath(os.path.expanduser(args.work_dir))
        self.work_dir_local = self.work_dir
        args.acoustic_setting = self.acoustic_setting
        return args

    def dir(self):
        work_dir = self.work_dir
        dirs = {"code": work_dir + "/SAR-SSL/code", "data": self.work_dir_local + "/data",
                "gerdata": self.work_dir_local + "/SAR-SSL/data", "exp": work_dir + "/SAR-SSL/exp"}
        dirs["micsig_simu_pretrain"] = dirs["gerdata"] + "/MicSig/simu/pretrain"
        dirs["micsig_simu_preval"] = dirs["gerdata"] + "/MicSig/simu/preval"
        dirs["micsig_simu_pretest"] = dirs["gerdata"] + "/MicSig/simu/pretest"
        dirs["micsig_simu_pretest_ins"] = [dirs["gerdata"] + "/MicSig/simu/pretest_ins_T1000"]
        dirs["log_pretrain"] = dirs["exp"] + "/pretrain/" + self.time
        return dirs


class opt_downstream():
    """Downstream (TDOA / DRR / C50 / T60 / ABS regression on the pretrained encoders) options, simulated-data branch.
    ``--ds-nepoch/--ds-num/--ds-lr-set/--ds-bs-set/--ds-eval-num`` are build-side overrides of the reference's hard-wired sweep
    (code/opt.py:197-209) so that a short run is possible; left unset, the reference's values apply."""

    def __init__(self):
        self.time = time.strftime("%m%d%H%M", time.localtime(time.time()))
        self.work_dir = os.path.abspath(os.path.expanduser(r"~"))
        self.work_dir_local = self.work_dir
        self.acoustic_setting = {"sound_speed": 343.0, "fs": 16000, "snr_range": [15, 30], "nmic": 2, "mic_dist_range": [0.03, 0.20]}
        self.extra_info, self.ds_token, self.ds_head, self.ds_embed, self.ds_nsimroom = "", "", "", "", 0
        self.simu_exp, self.ntrail = True, 1

    def parse(self, argv=None):
        p = argparse.ArgumentParser(description="Self-supervised learing for multi-channel audio processing")
        p.add_argument("--gpu-id", type=str, default="6,", metavar="GPU", help="GPU ID")
        p.add_argument("--workers", type=int, default=4, metavar="Worker", help="number of workers (default: 4)")
        p.add_argument("--no-cuda", action="store_true", default=False)
        p.add_argument("--use-amp", action="store_true", default=False, help="mixed precision (the 'hybrid' mode; SARSSL_AMP_DTYPE=fp16|bf16 selects the others)")
        p.add_argument("--seed", type=int, default=1, metavar="Seed")
        p.add_argument("--checkpoint-start", action="store_true", default=False)
        p.add_argument("--time", type=str, default=self.time, metavar="Time")
        p.add_argument("--work-dir", type=str, default=self.work_dir, metavar="WorkDir")
        p.add_argument("--sources", type=int, nargs="+", default=[1])
        p.add_argument("--source-state", type=str, default="static")
        p.add_argument("--simu-exp", action="store_true", default=False)
        p.add_argument("--ds-train", action="store_true", default=False)
        p.add_argument("--ds-trainmode", type=str, default="finetune")
        p.add_argument("--ds-task", type=str, nargs="+", default=["TDOA"])
        p.add_argument("--ds-token", type=str, default="all")
        p.add_argument("--ds-head", type=str, default="mlp")
        p.add_argument("--ds-embed", type=str, default="spat")
        p.add_argument("--ds-nsimroom", type=int, default=0)
        p.add_argument("--ds-real-sim-ratio", type=int, nargs="+", default=[1, 1])
        p.add_argument("--ds-test", action="store_true", default=False)
        p.add_argument("--test-mode", type=str, default="cal_metric_wo_info")
        p.add_argument("--ds-nepoch", type=int, default=None)
        p.add_argument("--ds-num", type=int, default=None)
        p.add_argument("--ds-lr-set", type=float, nargs="+", default=None)
        p.add_argument("--ds-bs-set", type=int, nargs="+", default=None)
        p.add_argument("--ds-eval-num", type=int, default=None, help="size of the val / test sets (reference: 1000, test_large 4000)")
        args = p.parse_args(argv)
        assert (args.ds_train + args.ds_test) == 1, "Downstream stage (train or test) is not defined"
        assert args.ds_trainmode in ["scratchLOW", "finetune", "lineareval"], "Downstream train mode in not defined"
        assert args.test_mode in ["cal_metric", "cal_metric_wo_info", "vis_embed"], "Test mode is undefined"
        self.simu_exp, self.time = args.simu_exp, args.time
        self.work_dir = os.path.abspath(os.path.expanduser(args.work_dir))
        self.work_dir_local = self.work_dir
        self.ds_token, self.ds_head, self.ds_embed, self.ds_nsimroom = args.ds_token, args.ds_head, args.ds_embed, args.ds_nsimroom
        args.ds_specifics = {"task": args.ds_task}
        args.acoustic_setting = self.acoustic_setting
        if not self.simu_exp:
            raise SystemExit("only the simulated-data branch (--simu-exp) of the downstream stage is implemented")
        bs_set = args.ds_bs_set or [8]
        lr_set = args.ds_lr_set or [0.001, 0.0005, 0.0001, 0.00005]
        nepoch = args.ds_nepoch or 200
        num = args.ds_num if args.ds_num is not None else args.ds_nsimroom * 100
        ntrial = int(np.maximum(1, round(32 / (args.ds_nsimroom + 10e-4))))
        self.ntrail = ntrial
        args.ds_setting = {t: {"nepoch": nepoch, "num": num, "lr_set": lr_set, "bs_set": bs_set, "ntrial": ntrial}
                           for t in ("TDOA", "DRR", "C50", "T60", "ABS")}
        self.extra_info = "R" + str(args.ds_nsimroom)
        return args

    def dir(self):
        work_dir = self.work_dir
        dirs = {"code": work_dir + "/SAR-SSL/code", "data": self.work_dir_local + "/data",
                "gerdata": self.work_dir_local + "/SAR-SSL/data", "exp": work_dir + "/SAR-SSL/exp"}
        dirs["micsig_train_simu"] = []
        train_dir = dirs["gerdata"] + "/MicSig/simu_ds/train"
        for trail_idx in range(self.ntrail):
            dirs["micsig_train_simu"] += [[os.path.join(train_dir, "R" + str(trail_idx * self.ds_nsimroom + r + 1))
                                           for r in range(self.ds_nsimroom)]]
        dirs["micsig_val_simu"] = dirs["gerdata"] + "/MicSig/simu_ds/val"
        dirs["micsig_test_simu"] = dirs["gerdata"] + "/MicSig/simu_ds/test"
        flag = "sim_"
        dirs["log_pretrain"] = dirs["exp"] + "/pretrain/" + self.time
        dirs["log_task"] = dirs["exp"] + "/TASK/" + self.time
        for mode, name in (("scratchLOW", "scratchlow"), ("finetune", "finetune"), ("lineareval", "lineareval")):
            dirs["log_task_" + mode] = (dirs["log_task"] + "/" + name + "-" + self.ds_token + "-" + self.ds_head + "-NUM-LR-BAS-TRI-"
                                        + self.ds_embed + "-" + flag + self.extra_info)
        return dirs
