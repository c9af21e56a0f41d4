"""Seeding, parameter counting and the cosine learning-rate schedule (code/common/utils.py:39-56, 59-72, 108-139)."""
import json
import random

import numpy as np
import torch

from ..runtime import RT


def set_seed(seed):
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    RT.manual_seed(seed)


set_random_seed = set_seed


def get_nparams(model, param_key_list=[]):
    nparam_sum = 0.0
    nparam = {k: 0 for k in param_key_list}
    for key, value in model.named_parameters():
        nparam_sum += value.numel() / 1000000
        for k in param_key_list:
            if k in key:
                nparam[k] += value.numel() / 1000000
    return nparam, nparam_sum


def create_learning_rate_schedule(total_steps, base, decay_type, warmup_steps, linear_end=1e-5):
    def step_fn(step):
        lr = base
        progress = np.clip((step - warmup_steps) / float(total_steps - warmup_steps), 0.0, 1.0)
        if decay_type == "linear":
            lr = linear_end + (lr - linear_end) * (1.0 - progress)
        elif decay_type == "cosine":
            lr = lr * 0.5 * (1.0 + np.cos(np.pi * progress))
        else:
            raise ValueError(f"Unknown lr type {decay_type}")
        if warmup_steps:
            lr = lr * np.minimum(1.0, step / warmup_steps)
        return np.asarray(lr, dtype=np.float32)
    return step_fn


def save_config_to_file(config, file_path):
    with open(file_path, "w") as f:
        json.dump(config, f, indent=4, default=str)
