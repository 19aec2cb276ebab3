"""train.py -- same contract as the reference's (train.py:1-8): load a YAML config, Trainer(config).train()."""
import sys

import yaml

from trainer import Trainer

if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else 'configs/basic_config.yaml'
    with open(path) as file:
        config = yaml.full_load(file)
    Trainer(config).train()
