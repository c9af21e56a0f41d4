"""Command-line options and directory layout of the pretraining entry point: same flag names / defaults as
``opt_pretrain`` in the reference (code/opt.py:6-115)."""
import argparse
import os
import time


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
        p.add_argument("--use-amp", action="store_true", default=False, help="Use mixed precision (bf16 on MI355X)")
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
        dirs["log_pretrain"] = dirs["exp"] + "/pretrain/" + self.time
        return dirs
