"""Minimal attribute-dict config node (yacs is not available on the target image).

Supports exactly what the hot path and its callers use from ``yacs.config.CfgNode``
in the reference (config.py:14-106, pose_estimators/config.py:10-56): attribute and
item access, nested nodes, ``clone()``, ``merge_from_file`` (YAML) and
``merge_from_list``.
"""
from __future__ import annotations

import copy


class CfgNode(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def _merge(self, other: dict, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f"non-existent config key: {path}{k}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise TypeError(f"{path}{k} is a config node")
                self[k]._merge(v, f"{path}{k}.")
            else:
                self[k] = list(v) if isinstance(v, (list, tuple)) else v

    def merge_from_file(self, path: str):
        import yaml

        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, items):
        assert len(items) % 2 == 0
        for k, v in zip(items[0::2], items[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"non-existent config key: {k}")
            node[parts[-1]] = v
