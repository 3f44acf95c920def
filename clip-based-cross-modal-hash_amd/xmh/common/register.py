"""Plugin registry with the API of the reference's ``common/register.py`` (:9-302): the same decorator and
lookup names, the same failure modes (AssertionError for a wrong base class :81-83/:161-163, KeyError for a
duplicate name :84-89, ``None`` for an unknown name :219-236) -- so a runner/model written against the
reference registers here unchanged.  One table per kind instead of one hand-written method per kind."""
from __future__ import annotations

from typing import Callable, Dict, Optional


class Registry:
    mapping: Dict[str, Dict[str, object]] = {
        "model_name_mapping": {},
        "runner_name_mapping": {},
        "optimizer_name_mapping": {},
        "tokenizer_name_mapping": {},
        "dataset_name_mapping": {},
        "state": {},
        "paths": {},
    }

    # ---- registration ---------------------------------------------------------------------------
    @classmethod
    def _register(cls, kind: str, name: str, base_of: Optional[Callable[[], type]] = None):
        table = cls.mapping[kind + "_name_mapping"]

        def wrap(obj):
            if base_of is not None:
                base = base_of()
                assert issubclass(obj, base), "All %ss must inherit %s class" % (kind, base.__name__)
            if name in table:
                raise KeyError("Name '{}' already registered for {}.".format(name, table[name]))
            table[name] = obj
            return obj
        return wrap

    @classmethod
    def register_model(cls, name):
        def base():
            from ..models.base import BaseModel
            return BaseModel
        return cls._register("model", name, base)

    @classmethod
    def register_runner(cls, name):
        def base():
            from ..runners.base import BaseTrainer
            return BaseTrainer
        return cls._register("runner", name, base)

    @classmethod
    def register_optimizer(cls, name):
        return cls._register("optimizer", name)

    @classmethod
    def register_tokenizer(cls, name):
        return cls._register("tokenizer", name)

    @classmethod
    def register_dataset(cls, name):
        return cls._register("dataset", name)

    @classmethod
    def register_path(cls, name, path):
        assert isinstance(path, str), "All path must be str."
        if name in cls.mapping["paths"]:
            raise KeyError("Name '{}' already registered.".format(name))
        cls.mapping["paths"][name] = path

    @classmethod
    def register(cls, name, obj):
        cur = cls.mapping["state"]
        parts = name.split(".")
        for part in parts[:-1]:
            cur = cur.setdefault(part, {})
        cur[parts[-1]] = obj

    # ---- lookup ---------------------------------------------------------------------------------
    @classmethod
    def _get(cls, kind, name):
        return cls.mapping[kind + "_name_mapping"].get(name, None)

    @classmethod
    def get_model_class(cls, name):
        return cls._get("model", name)

    @classmethod
    def get_runner_class(cls, name):
        return cls._get("runner", name)

    @classmethod
    def get_optimizer_class(cls, name):
        return cls._get("optimizer", name)

    @classmethod
    def get_tokenizer_class(cls, name):
        return cls._get("tokenizer", name)

    @classmethod
    def get_dataset_class(cls, name):
        return cls._get("dataset", name)

    @classmethod
    def list_models(cls):
        return sorted(cls.mapping["model_name_mapping"].keys())

    @classmethod
    def list_runners(cls):
        return sorted(cls.mapping["runner_name_mapping"].keys())

    @classmethod
    def list_optimizers(cls):
        return sorted(cls.mapping["optimizer_name_mapping"].keys())

    list_optimizer = list_optimizers          # the reference spells it in the singular (common/register.py:251)

    @classmethod
    def list_tokenizers(cls):
        return sorted(cls.mapping["tokenizer_name_mapping"].keys())

    @classmethod
    def list_datasets(cls):
        return sorted(cls.mapping["dataset_name_mapping"].keys())

    @classmethod
    def get_path(cls, name):
        return cls.mapping["paths"].get(name, None)

    @classmethod
    def get(cls, name, default=None, no_warning=False):
        value = cls.mapping["state"]
        for part in name.split("."):
            value = value.get(part, default) if isinstance(value, dict) else default
            if value is default:
                break
        return value

    @classmethod
    def unregister(cls, name):
        return cls.mapping["state"].pop(name, None)


registry = Registry()
