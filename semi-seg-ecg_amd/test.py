"""Evaluation entry point with the reference's CLI (``src/test.py:14-80``):
``python test.py -f base.yaml [-o bench.yaml] [--output_dir ..] [--exp_name ..] [--model_path ckpt.pth]``;
dispatches to ``algorithms.<algorithm>.test(config)``, which reloads ``best-<target_metric>.pth`` (or ``--model_path``),
evaluates the test split on the device and writes ``test_metrics.csv``, ``test_outputs.npy``, ``test_labels.npy``."""
import argparse
import os

import yaml

import algorithms
from train import deep_merge


def parse() -> dict:
    parser = argparse.ArgumentParser('SemiSegECG testing on the MI355X hot path')
    parser.add_argument('-f', '--config_path', dest='config_path', required=True, type=str, metavar='FILE',
                        help='YAML config file path')
    parser.add_argument('-o', '--override_config_path', dest='override_config_path', default=None, type=str, metavar='FILE',
                        help='YAML config file path to override')
    parser.add_argument('--output_dir', default="", type=str, metavar='DIR', help='path where to save')
    parser.add_argument('--exp_name', default="", type=str, help='experiment name')
    parser.add_argument('--model_path', default="", type=str, metavar='PATH', help='saved from checkpoint')
    args = parser.parse_args()
    with open(os.path.realpath(args.config_path), 'r') as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    if args.override_config_path:
        with open(os.path.realpath(args.override_config_path), 'r') as f:
            config = deep_merge(config, yaml.load(f, Loader=yaml.FullLoader))
    for k, v in vars(args).items():
        if v:
            if k == 'model_path':
                config.setdefault('test', {})
                config['test'] = dict(config['test'] or {}, model_path=v)
            else:
                config[k] = v
    return config


def main(config):
    name = config.get('algorithm')
    if name not in algorithms.__dict__ or not hasattr(algorithms.__dict__[name], 'test'):
        raise ValueError(f"Invalid algorithm: {name}")
    return algorithms.__dict__[name].test(config)


if __name__ == "__main__":
    main(parse())
