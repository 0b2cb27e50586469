"""Minimal stand-in for the `gensim` package (absent from this image).

Test-harness code only: lets tools/make_golden.py import the reference's
pure-Python modules in this container.  The reference only ever uses
``len(dictionary)`` on the object returned by ``Dictionary.from_corpus``.
Contains no reference code.
"""
from . import utils  # noqa: F401
