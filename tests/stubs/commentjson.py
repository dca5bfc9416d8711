"""Test stub: the reference driver imports `commentjson` only for json-with-comments loading; the test configs have no comments."""
from json import load, loads, dump, dumps  # noqa
