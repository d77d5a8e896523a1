"""Attribute-access config dict accepted by ``get_model(config)`` (stands in for OmegaConf's DictConfig, which the
reference uses at ``cinema/mae/mae.py:231-282``; hydra/omegaconf are not required here)."""

from __future__ import annotations


class Config(dict):
    def __getattr__(self, key: str):  # noqa: ANN204
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key: str, value) -> None:  # noqa: ANN001
        self[key] = value


def to_config(obj):  # noqa: ANN001, ANN201
    if isinstance(obj, dict):
        return Config({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_config(v) for v in obj]
    return obj
