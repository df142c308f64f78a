"""Dataset definition files: the YAML description of where a dataset's images and labels live.

Host-side mirror of yogo/data/dataset_definition_file.py:97-494 and yogo/data/split_fractions.py -- same class names, fields,
methods and exceptions, so callers written against the reference keep working:

* a definition file lists *literal* specifications (``image_path`` + ``label_path``) and *recursive* ones (``defn_path`` to
  another definition file, relative to the including file); loading flattens the tree into sets of literal specifications;
* the tree must be a tree: a definition file that is reached twice on one path (a cycle) and a literal specification that is
  reached twice (a duplicate) are rejected with ``InvalidDatasetDefinitionFile``; so are files whose ``class_names`` differ;
* with a ``test_paths`` section the two sections are loaded separately (and must be disjoint); without one, ``test_paths`` of
  included files are folded into the dataset paths;
* every literal specification must point at existing directories with at least one label file (``FileNotFoundError``).

The reference parses with ruamel.yaml (YAML 1.2, safe); this build has PyYAML (safe loader), which reads these files the same.
Pinned by the reference's own known-answer tests and fixture files (tests/test_dataset_definition.py, tests/golden/dataset_defns/).
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass
from enum import Enum
from pathlib import Path
from typing import Any, Dict, FrozenSet, List, Optional, Set, Tuple, Union

import yaml


class InvalidSplitFraction(Exception):
    pass


class SplitFractions:
    """yogo/data/split_fractions.py:8-104: fractions of the data used for training, validation and testing."""

    def __init__(self, train: float, val: float, test: Optional[float]) -> None:
        self.train, self.val, self.test = train, val, test
        parts = (self.train, self.val, self.test or 0)
        if not all(0 <= v <= 1 for v in parts):
            raise ValueError(f"train, val, and test must be in range [0,1]; they are {self.train}, {self.val}, and {self.test}")
        if not abs(sum(parts) - 1) < 1e-10:
            raise ValueError(f"train, val, and test must sum to 1; they sum to {sum(parts)}")

    def __repr__(self) -> str:
        return f"SplitFractions(train={self.train}, val={self.val}, test={self.test})"

    def __contains__(self, item: object) -> bool:
        return item in self.to_dict()

    def __eq__(self, other: object) -> bool:
        return isinstance(other, SplitFractions) and (self.train, self.val, self.test) == (other.train, other.val, other.test)

    @classmethod
    def train_only(cls) -> "SplitFractions":
        return cls(1, 0, None)

    @classmethod
    def from_list(cls, lst: List[float], test_paths_present: bool = True) -> "SplitFractions":
        if len(lst) != 3:
            raise InvalidSplitFraction(f"SplitFractions.from_list's list must have length 3, but found length {len(lst)}")
        return cls.from_dict(dict(zip(["train", "val", "test"], lst)), test_paths_present=test_paths_present)

    @classmethod
    def from_dict(cls, dct: Dict[str, float], test_paths_present: bool = True) -> "SplitFractions":
        if test_paths_present and "test" in dct:
            raise InvalidSplitFraction(
                "when `test_paths` is present in a dataset descriptor file, 'test' is not a valid key for "
                "`dataset_split_fractions`, since we will use all the data from `test_paths` for testing")
        if not any(k in dct for k in ("train", "val", "test")):
            raise InvalidSplitFraction(f"dct must have keys `train`, `val`, and `test` - found keys {dct.keys()}")
        if len(dct) > 3:
            raise InvalidSplitFraction(f"dct must have keys `train`, `val`, and `test` only, but found {len(dct)} keys")
        return cls(dct["train"], dct["val"], dct.get("test", None))

    def to_dict(self) -> Dict[str, float]:
        return {k: v for k, v in (("train", self.train), ("val", self.val), ("test", self.test)) if v is not None}

    def keys(self) -> List[str]:
        return list(self.to_dict().keys())

    def partition_sizes(self, total_size: int) -> Dict[str, int]:
        """split sizes that add up to total_size: every split but the last is rounded, the last takes the remainder"""
        fractions = self.to_dict()
        names = self.keys()
        sizes = {k: round(fractions[k] * total_size) for k in names[:-1]}
        sizes[names[-1]] = total_size - sum(sizes.values())
        if any(sz < 0 for sz in sizes.values()) or sum(sizes.values()) != total_size:
            raise ValueError(f"could not create valid dataset split sizes: {sizes}, full dataset size is {total_size}")
        return sizes


class InvalidDatasetDefinitionFile(Exception): ...


@dataclass
class LiteralSpecification:
    """an (image directory, label directory) pair -- dataset_definition_file.py:100-152"""

    image_path: Path
    label_path: Path

    @classmethod
    def from_dict(cls, dct: Dict[str, str]) -> "LiteralSpecification":
        if len(dct) != 2:
            raise InvalidDatasetDefinitionFile(f"LiteralSpecification must have two keys; found {len(dct)}")
        if "image_path" not in dct or "label_path" not in dct:
            hint = " ('defn_path' found: a recursive specification was handed to the literal parser)" if "defn_path" in dct else ""
            raise InvalidDatasetDefinitionFile("LiteralSpecification must have keys 'image_path' and 'label_path'" + hint)
        return cls(Path(dct["image_path"]), Path(dct["label_path"]))

    def to_dict(self) -> Dict[str, str]:
        return {"image_path": str(self.image_path), "label_path": str(self.label_path)}

    def __eq__(self, other: object) -> bool:
        return isinstance(other, LiteralSpecification) and (self.image_path, self.label_path) == (other.image_path, other.label_path)

    def __hash__(self) -> int:
        return hash((self.image_path, self.label_path))


class SpecificationsKey(Enum):
    DATASET_PATHS = "dataset_paths"
    TEST_DATASET_PATHS = "test_paths"
    ALL_DATASET_PATHS = "all_paths"


def _read_yaml(path: Path) -> Dict[str, Any]:
    with open(path, "r") as f:
        return yaml.safe_load(f)


def _classes_of(data: Dict[str, Any]) -> List[str]:
    try:
        return data["class_names"]
    except KeyError as e:
        raise InvalidDatasetDefinitionFile("`classes` is a required key in the dataset definition file") from e


@dataclass
class DatasetDefinition:
    """The flattened definition -- dataset_definition_file.py:161-249."""

    _dataset_paths: Set[LiteralSpecification]
    _test_dataset_paths: Set[LiteralSpecification]
    classes: List[str]
    thumbnail_augmentation: Optional[Dict[str, Union[Path, List[Path]]]]
    split_fractions: SplitFractions

    @property
    def dataset_paths(self) -> List[LiteralSpecification]:
        return list(self._dataset_paths)

    @property
    def test_dataset_paths(self) -> List[LiteralSpecification]:
        return list(self._test_dataset_paths)

    @property
    def all_dataset_paths(self) -> List[LiteralSpecification]:
        return list(self._dataset_paths | self._test_dataset_paths)

    @classmethod
    def from_yaml(cls, path: Union[str, Path]) -> "DatasetDefinition":
        path = Path(path)
        data = _read_yaml(path)
        has_tests = "test_paths" in data
        classes = _classes_of(data)
        if has_tests:
            train_specs = cls._load_dataset_specifications(path, classes, dataset_paths_key=SpecificationsKey.DATASET_PATHS)
            test_specs = cls._load_dataset_specifications(path, classes, exclude_ymls=[path], exclude_specs=train_specs,
                                                          dataset_paths_key=SpecificationsKey.TEST_DATASET_PATHS)
        else:
            train_specs = cls._load_dataset_specifications(path, classes, dataset_paths_key=SpecificationsKey.ALL_DATASET_PATHS)
            test_specs = set()
        train_specs = cls._check_dataset_paths(train_specs)
        test_specs = cls._check_dataset_paths(test_specs)
        if "dataset_split_fractions" in data:
            fractions = SplitFractions.from_dict(data["dataset_split_fractions"], test_paths_present=has_tests)
        else:
            fractions = SplitFractions.train_only()
        return cls(_dataset_paths=train_specs, _test_dataset_paths=test_specs, classes=classes,
                   thumbnail_augmentation=cls._load_thumbnails(classes, data), split_fractions=fractions)

    def __add__(self, other: "DatasetDefinition") -> "DatasetDefinition":
        for what, a, b in (("classes", self.classes, other.classes),
                           ("thumbnail augmentation", self.thumbnail_augmentation, other.thumbnail_augmentation),
                           ("split fractions", self.split_fractions, other.split_fractions)):
            if a != b:
                raise ValueError(f"cannot concatenate two dataset definitions with different {what}")
        return DatasetDefinition(_dataset_paths=self._dataset_paths | other._dataset_paths,
                                 _test_dataset_paths=self._test_dataset_paths | other._test_dataset_paths, classes=self.classes,
                                 thumbnail_augmentation=self.thumbnail_augmentation, split_fractions=self.split_fractions)

    def __eq__(self, other: object) -> bool:
        return isinstance(other, DatasetDefinition) and (
            self._dataset_paths == other._dataset_paths and self._test_dataset_paths == other._test_dataset_paths
            and self.classes == other.classes and self.thumbnail_augmentation == other.thumbnail_augmentation
            and self.split_fractions == other.split_fractions)

    # ---- loading ---------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _extract_specs(yml_path: Path, dataset_paths_key: SpecificationsKey) -> Tuple[List[str], List[Dict[str, str]]]:
        """(class names, specification dicts of the requested section) of one file -- :294-323"""
        data = _read_yaml(yml_path)
        classes = _classes_of(data)
        if dataset_paths_key == SpecificationsKey.ALL_DATASET_PATHS:
            specs = [s for key in (SpecificationsKey.DATASET_PATHS, SpecificationsKey.TEST_DATASET_PATHS)
                     for s in data.get(key.value, dict()).values()]
        else:
            specs = list(data.get(dataset_paths_key.value, dict()).values())
        return classes, specs

    @staticmethod
    def _load_dataset_specifications(yml_path: Path, classes: List[str], exclude_ymls: List[Path] = [],
                                     exclude_specs: Union[Set[LiteralSpecification], FrozenSet[LiteralSpecification]] = frozenset(),
                                     dataset_paths_key: SpecificationsKey = SpecificationsKey.DATASET_PATHS) -> Set[LiteralSpecification]:
        """Flatten one section of a definition file into literal specifications -- :325-412.  exclude_ymls = the definition
        files on the path from the root (reaching one again is a cycle); exclude_specs = literal specifications that must not
        show up (the training set while the test set is loaded)."""
        found: Set[LiteralSpecification] = set()
        file_classes, specs = DatasetDefinition._extract_specs(yml_path, dataset_paths_key)
        if file_classes != classes:
            raise InvalidDatasetDefinitionFile(f"classes mismatch in {yml_path}")
        for spec in specs:
            if "defn_path" in spec:
                child = Path(spec["defn_path"])
                if not child.is_absolute():
                    child = yml_path.parent / child   # relative to the including file
                if child in exclude_ymls:
                    raise InvalidDatasetDefinitionFile(f"cycle found: {spec['defn_path']} is duplicated")
                child_specs = DatasetDefinition._load_dataset_specifications(
                    child, classes, exclude_ymls=[child, *exclude_ymls], dataset_paths_key=dataset_paths_key)
                if "classes" in spec and spec["classes"] != classes:
                    raise InvalidDatasetDefinitionFile(f"classes mismatch in {spec['defn_path']}")
                DatasetDefinition._check_for_non_disjoint_sets(found, child_specs)
                found |= child_specs
            elif "image_path" in spec and "label_path" in spec:
                literal = LiteralSpecification.from_dict(spec)
                DatasetDefinition._check_for_non_disjoint_sets(found, {literal})
                found.add(literal)
            else:
                raise InvalidDatasetDefinitionFile(f"Invalid spec in dataset_paths: {spec}")
        duplicates = found & exclude_specs
        if duplicates:
            raise InvalidDatasetDefinitionFile(f"duplicate literal definition found in exclude paths!\nduplicates are: {duplicates}")
        return found

    @staticmethod
    def _check_for_non_disjoint_sets(s1: Set, s2: Set) -> None:
        common = s1 & s2
        if common:
            raise InvalidDatasetDefinitionFile(f"duplicates found when trying to add s1 to s2\nduplicates are: {common}")

    @staticmethod
    def _load_thumbnails(classes: List[str], yaml_data: Dict[str, Any]) -> Optional[Dict[str, Union[Path, List[Path]]]]:
        """`thumbnail_augmentation: {class name: directory | [directories]}` -- :424-447"""
        if "thumbnail_augmentation" not in yaml_data:
            return None
        table = yaml_data["thumbnail_augmentation"]
        if not isinstance(table, dict):
            raise InvalidDatasetDefinitionFile("thumbnail_augmentation must map class names to paths to thumbnail directories "
                                               "(e.g. `misc: /path/to/thumbnails/misc`)")
        for name in table:
            if name not in classes:
                raise InvalidDatasetDefinitionFile(f"thumbnail_augmentation class {name} is not a valid class name")
        for name, value in table.items():
            if not isinstance(value, list):
                table[name] = [Path(value)]
        return table

    @staticmethod
    def _check_dataset_paths(dataset_paths: Set[LiteralSpecification], prune: bool = False) -> Set[LiteralSpecification]:
        """every specification needs an image directory and a non-empty label directory -- :449-476"""
        missing: Set[LiteralSpecification] = set()
        for spec in dataset_paths:
            if spec.image_path.is_dir() and spec.label_path.is_dir() and any(True for _ in spec.label_path.iterdir()):
                continue
            message = ("image_path or label_path do not lead to a directory, or there are no labels.\n"
                       f"image_path={spec.image_path}\nlabel_path={spec.label_path}\n")
            if not prune:
                raise FileNotFoundError(message)
            warnings.warn(message + "will prune.")
            missing.add(spec)
        return dataset_paths - missing

    @staticmethod
    def _extract_dataset_paths(path: Path) -> List[Dict[str, str]]:
        """the raw specification dicts of a file's `dataset_paths` section -- :478-494"""
        data = _read_yaml(path)
        if "dataset_paths" not in data:
            raise InvalidDatasetDefinitionFile(f"Missing dataset_paths for definition file at {path}")
        return list(data["dataset_paths"].values())
