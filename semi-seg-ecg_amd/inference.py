"""Inference entry point with the reference's CLI (``src/inference.py:16-131``):
``python inference.py -f base.yaml [-o override.yaml] [--output_dir ..] [--exp_name ..] [--model_path ckpt.pth]``.

Reloads ``test.model_path`` (or ``<output_dir>/<exp_name>/best-<target_metric>.pth``), drops ``auxiliary_head*`` entries from the
checkpoint as the reference does, runs the eval-mode forward (BN-folded HIP kernels) over the test split and writes the softmax
probabilities ``(records, classes, L)`` to ``test_outputs.npy``.  The softmax comes from ``ssecg_softmax_conf_argmax`` (one pass:
probabilities + arg-max); no labels are needed, as in the reference.  ``test.use_amp`` (default false, src/inference.py:110)
selects the 16-bit eval path (``ssecg.amp.eval_autocast``), as the reference's forward runs inside autocast then."""
import os

import numpy as np
import torch

from algorithms.base import init_model_from_cfg
from ssecg import amp as SAMP
from ssecg import functional as SF
from ssecg import ops
from utils.semi_dataset import build_seg_dataset, get_dataloader


@torch.no_grad()
def inference(config):
    output_dir = os.path.join(config['output_dir'], config['exp_name'])
    os.makedirs(output_dir, exist_ok=True)
    device = torch.device(config['device'])
    dataset_test = build_seg_dataset(config['dataset'], split="test")
    loader = get_dataloader(dataset_test, is_distributed=False, mode='test', **config['dataloader'])
    model = init_model_from_cfg(config, train=False)
    tcfg = config.get('test') or {}
    if tcfg.get('model_path'):
        checkpoint_path = tcfg['model_path']
    else:
        checkpoint_path = os.path.join(output_dir, f"best-{tcfg.get('target_metric', 'loss')}.pth")
    assert os.path.exists(checkpoint_path), f"Checkpoint not found: {checkpoint_path}"
    state_dict = torch.load(checkpoint_path, map_location='cpu', weights_only=False)['model']
    for k in list(state_dict.keys()):          # drop the auxiliary head (src/inference.py:100-103)
        if k.startswith('auxiliary_head'):
            del state_dict[k]
    print(model.load_state_dict(state_dict))
    model.to(device)
    model.eval()
    chunks = []
    for samples in loader:
        inputs = samples['ecg'].to(device, non_blocking=True)
        # inside autocast(enabled=config['test'].get('use_amp', False)) in the reference (src/inference.py:110-117): the 16-bit eval path
        # when that key is set, never K-split (a record's probabilities do not depend on the batch it shares)
        with SAMP.eval_autocast(model, bool(tcfg.get('use_amp', False))), ops.ksplit_disabled():
            logits = model(inputs, return_loss=False)['seg_logits']
        chunks.append(SF.pseudo_label(logits, want_prob=True)[2].cpu())
    outputs = torch.cat(chunks, dim=0).numpy()
    np.save(os.path.join(output_dir, 'test_outputs.npy'), outputs)
    print("Done!")
    return outputs


if __name__ == "__main__":
    from test import parse      # the same flags as test.py (src/inference.py:16-73 == src/test.py:14-72)
    inference(parse())
