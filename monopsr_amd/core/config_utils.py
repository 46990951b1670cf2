"""YAML -> nested attribute objects, as the reference's core/config_utils.py:8-85 does (duplicate keys rejected)."""
import os

import yaml


class ConfigObj:
    def __init__(self, d):
        for k, v in d.items():
            setattr(self, k, ConfigObj(v) if isinstance(v, dict) else v)

    def __contains__(self, k):
        return k in self.__dict__

    def get(self, k, default=None):
        return self.__dict__.get(k, default)


class _UniqueKeyLoader(yaml.SafeLoader):
    def construct_mapping(self, node, deep=False):
        seen = set()
        for key_node, _ in node.value:
            key = self.construct_object(key_node, deep=deep)
            if key in seen:
                raise ValueError("Duplicate key {!r} in config".format(key))
            seen.add(key)
        return super().construct_mapping(node, deep)


def parse_yaml_config(yaml_path):
    with open(yaml_path, "r") as f:
        return ConfigObj(yaml.load(f, Loader=_UniqueKeyLoader))


def default_config():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return parse_yaml_config(os.path.join(here, "configs", "monopsr_model_000.yaml"))
