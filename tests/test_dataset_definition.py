"""Dataset definition files: the reference's own known answers (its tests/test_dataset_definition.py, restated) on the
reference's own fixture files (tests/golden/dataset_defns/*.yml = its tests/fake-data/defns, copied as data).  The fixture
paths are relative to the working directory ("tests/fake-data/data/images1"), so each test runs inside a scratch tree that
holds the (empty) image / label files the reference's fake-data directory holds."""
import shutil
from pathlib import Path

import pytest

from yogo_amd.dataset_definition_file import (DatasetDefinition, InvalidDatasetDefinitionFile, InvalidSplitFraction,
                                              LiteralSpecification, SplitFractions)

FIXTURES = Path(__file__).parent / "golden" / "dataset_defns"


@pytest.fixture()
def defns(tmp_path, monkeypatch):
    for k in (1, 2, 3):
        for kind, ext in (("images", "png"), ("labels", "txt")):
            d = tmp_path / "tests" / "fake-data" / "data" / f"{kind}{k}"
            d.mkdir(parents=True)
            for j in (1, 2, 3):
                (d / f"img_{j}.{ext}").touch()
    out = tmp_path / "tests" / "fake-data" / "defns"
    shutil.copytree(FIXTURES, out)
    monkeypatch.chdir(tmp_path)
    return out


def load(defns, name):
    return DatasetDefinition.from_yaml(defns / name)


def test_literal_files_load(defns):
    for name in ("literal_1.yml", "literal_2.yml", "literal_3.yml"):
        d = load(defns, name)
        assert len(d.dataset_paths) == 1 and len(d.test_dataset_paths) == 0
        assert d.classes == ["you", "only", "glance", "once"]
    d = load(defns, "literal_123.yml")
    assert len(d.dataset_paths) == 3 and len(d.test_dataset_paths) == 0
    assert LiteralSpecification(Path("tests/fake-data/data/images2"), Path("tests/fake-data/data/labels2")) in d._dataset_paths
    assert d.split_fractions == SplitFractions(1, 0, 0)


def test_recursive_files_flatten(defns):
    assert load(defns, "recursive_1.yml") == load(defns, "literal_1.yml")
    both = load(defns, "literal_1.yml") + load(defns, "literal_2.yml")
    assert load(defns, "recursive_1_literal_2.yml") == both
    assert load(defns, "recursive_1_literal_2.yml") == load(defns, "recursive_2_literal_1.yml")   # order does not matter
    assert load(defns, "recursive_rec_123.yml") == load(defns, "literal_123.yml")                 # two levels deep
    assert load(defns, "recursive_123.yml") == load(defns, "literal_123.yml")
    assert load(defns, "recursive_12.yml") == load(defns, "literal_12.yml")


@pytest.mark.parametrize("name", ["cycle_1.yml", "cycle_2.yml", "cycle_3.yml", "cycle_self.yml"])
def test_cycles_are_rejected(defns, name):
    with pytest.raises(InvalidDatasetDefinitionFile, match="cycle found"):
        load(defns, name)


def test_duplicates_missing_paths_and_class_mismatch(defns):
    with pytest.raises(InvalidDatasetDefinitionFile, match="duplicates"):
        load(defns, "duplicate_paths.yml")
    with pytest.raises(FileNotFoundError):
        load(defns, "literal-non-existant.yml")
    with pytest.raises(InvalidDatasetDefinitionFile, match="classes mismatch"):
        load(defns, "recursive_class_mismatch.yml")
    # an empty label directory counts as missing
    for f in (Path("tests/fake-data/data/labels1")).iterdir():
        f.unlink()
    with pytest.raises(FileNotFoundError):
        load(defns, "literal_1.yml")


def test_test_paths(defns):
    with_tests = load(defns, "literal_tests_123.yml")
    assert with_tests._dataset_paths == load(defns, "literal_12.yml")._dataset_paths
    assert with_tests._test_dataset_paths == load(defns, "literal_3.yml")._dataset_paths
    assert with_tests.split_fractions == SplitFractions(1, 0, None)
    rec = load(defns, "recursive_w_test.yml")
    assert rec._dataset_paths == with_tests._dataset_paths and rec._test_dataset_paths == with_tests._test_dataset_paths
    # a parent without test_paths folds its children's test_paths into the dataset paths
    folded = load(defns, "recursive_w_no_test.yml")
    assert folded._dataset_paths == with_tests._dataset_paths | with_tests._test_dataset_paths
    assert folded._test_dataset_paths == set()
    assert sorted(map(str, (s.image_path for s in with_tests.all_dataset_paths))) == [f"tests/fake-data/data/images{k}" for k in (1, 2, 3)]


def test_missing_split_fractions_mean_train_only(defns):
    assert load(defns, "no_split.yml").split_fractions == SplitFractions(train=1, val=0, test=None)
    assert load(defns, "no_split_no_test.yml").split_fractions == SplitFractions(train=1, val=0, test=None)


def test_concatenation_and_thumbnails(defns, tmp_path):
    a, b = load(defns, "literal_1.yml"), load(defns, "literal_tests_123.yml")
    with pytest.raises(ValueError, match="split fractions"):
        a + b
    text = (defns / "literal_1.yml").read_text()
    (defns / "thumbs.yml").write_text(text + "thumbnail_augmentation:\n  glance: /some/dir\n  once:\n    - /a\n    - /b\n")
    t = load(defns, "thumbs.yml")
    assert t.thumbnail_augmentation == {"glance": [Path("/some/dir")], "once": ["/a", "/b"]}
    with pytest.raises(ValueError, match="thumbnail augmentation"):
        a + t
    (defns / "thumbs_bad.yml").write_text(text + "thumbnail_augmentation:\n  twice: /some/dir\n")
    with pytest.raises(InvalidDatasetDefinitionFile, match="not a valid class name"):
        load(defns, "thumbs_bad.yml")
    (defns / "no_classes.yml").write_text("dataset_paths:\n  a:\n    image_path: x\n    label_path: y\n")
    with pytest.raises(InvalidDatasetDefinitionFile, match="required key"):
        load(defns, "no_classes.yml")
    (defns / "bad_spec.yml").write_text(text.replace("    label_path: tests/fake-data/data/labels1\n", ""))
    with pytest.raises(InvalidDatasetDefinitionFile, match="Invalid spec"):
        load(defns, "bad_spec.yml")
    assert LiteralSpecification.from_dict({"image_path": "i", "label_path": "l"}).to_dict() == {"image_path": "i", "label_path": "l"}
    with pytest.raises(InvalidDatasetDefinitionFile, match="two keys"):
        LiteralSpecification.from_dict({"image_path": "i", "label_path": "l", "x": "y"})


def test_split_fractions():
    s = SplitFractions.from_list([0.7, 0.2, 0.1], test_paths_present=False)
    assert (s.train, s.val, s.test) == (0.7, 0.2, 0.1) and s.keys() == ["train", "val", "test"] and "val" in s
    assert s.partition_sizes(101) == {"train": 71, "val": 20, "test": 10}
    assert SplitFractions.train_only().to_dict() == {"train": 1, "val": 0}
    assert SplitFractions.train_only().partition_sizes(9) == {"train": 9, "val": 0}
    assert repr(SplitFractions(0.5, 0.5, None)) == "SplitFractions(train=0.5, val=0.5, test=None)"
    with pytest.raises(InvalidSplitFraction, match="not a valid key"):
        SplitFractions.from_dict({"train": 0.5, "val": 0.4, "test": 0.1}, test_paths_present=True)
    with pytest.raises(InvalidSplitFraction, match="length 3"):
        SplitFractions.from_list([1.0, 0.0])
    with pytest.raises(ValueError, match="sum to 1"):
        SplitFractions(0.5, 0.2, 0.1)
    with pytest.raises(ValueError, match="in range"):
        SplitFractions(1.5, -0.5, None)
