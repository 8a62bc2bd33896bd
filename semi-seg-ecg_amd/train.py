"""Config-driven entry point with the reference's CLI (``src/train.py:14-90``):
``python train.py --config_path base.yaml [--override_config_path bench.yaml] [--output_dir ..] [--exp_name ..]
[--resume ..] [--start_epoch N]`` (short flags ``-f`` / ``-o`` as in the reference); the YAML's ``algorithm`` key selects the
plugin module; a ``test:`` section runs ``algo.test(config)`` on the main process afterwards (``src/train.py:86-90``)."""
import argparse

import torch.distributed as dist
import yaml

import algorithms


def deep_merge(base: dict, override: dict) -> dict:
    for k, v in override.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            deep_merge(base[k], v)
        else:
            base[k] = v
    return base


def parse() -> dict:
    parser = argparse.ArgumentParser('SemiSegECG training on the MI355X hot path')
    parser.add_argument('-f', '--config_path', dest='config_path', required=True, type=str, metavar='FILE',
                        help='YAML config file path')
    parser.add_argument('-o', '--override_config_path', dest='override_config_path', default=None, type=str, metavar='FILE',
                        help='YAML config file path to override')
    parser.add_argument('--output_dir', default="", type=str, metavar='DIR')
    parser.add_argument('--exp_name', default="", type=str)
    parser.add_argument('--resume', default="", type=str, metavar='PATH')
    parser.add_argument('--start_epoch', default=0, type=int, metavar='N')
    args = parser.parse_args()
    with open(args.config_path, 'r') as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    if args.override_config_path:
        with open(args.override_config_path, 'r') as f:
            config = deep_merge(config, yaml.load(f, Loader=yaml.FullLoader))
    for k, v in vars(args).items():
        if v:
            config[k] = v
    return config


def main(config):
    name = config['algorithm']
    if name not in algorithms.__dict__ or not hasattr(algorithms.__dict__[name], 'train'):
        raise ValueError(f"Unsupported algorithm on the MI355X hot path: {name} (available: base, fixmatch, mean_teacher, cps, stpp)")
    algo = algorithms.__dict__[name]
    algo.train(config)
    main_proc = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
    if dist.is_available() and dist.is_initialized():
        dist.barrier()   # rank 0's last checkpoint is on disk before anyone leaves
        dist.destroy_process_group()
    if config.get('test', False) and main_proc:   # src/train.py:86-90: evaluate the best checkpoint on the test split
        algo.test(config)


if __name__ == "__main__":
    main(parse())
