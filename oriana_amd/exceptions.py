# -*- coding: utf-8 -*-
"""Exception types of the reference surface (reference oriana/exceptions.py:6-11)."""

__all__ = ['IncompatibleShapeException', 'DatatypeException']


class IncompatibleShapeException(Exception):
    """Raised for a malformed dimension relation string."""


class DatatypeException(Exception):
    """Raised when a count matrix is built from an unsupported type."""
