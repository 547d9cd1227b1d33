"""Exceptions with the reference's names (cornac/exception.py:19-26)."""


class CornacException(Exception):
    """Exception base class to extend from"""


class ScoreException(CornacException):
    """Exception raised in score function when facing unknowns"""
