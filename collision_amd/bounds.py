"""Scene bounds of the sphere centres: per-axis (min, max) rows.  Public names and constructor
signatures of the reference's bounds module (collision/bounds.py:4-15); the classes themselves
come from reduce.specialise."""
from .reduce import specialise

# accumulator list of collision/bounds.py:5: start at +inf / -inf, fold with min / max
BoundsProgram, Bounds = specialise("BoundsProgram", "Bounds", [("INFINITY", "min"), ("-INFINITY", "max")],
                                   default_dtype=("float32", 3), dtype_keyword="coord_dtype")
BoundsProgram.__module__ = Bounds.__module__ = __name__
